#!/usr/bin/env python3
"""images/s of the int8 YOLO-face 56x56 forward on N MI355X (BASELINE.json metric).

A step = one pass of the hot path over this rank's batch of 4096 synthetic frames already resident in HBM:
fused forward kernel -> GPU box decode -> (N > 1) RCCL all-gather of the int8 heads.  Weak scaling: every rank
owns 4096 frames (BASELINE configs[1] at N=1, configs[2] = 32768 frames at N=8).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_GPU = 4096
ALGO_BYTES_PER_FRAME = 9408 + 882            # input + head, everything else stays in LDS
DENSE_OPS_PER_FRAME = 2 * 813792             # int8 ops eligible for MFMA (dense convs), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0                   # dense int8 MFMA = 2x the ~2.5 PF bf16 rate (MI355X_MICROARCH.md, Matrix cores)


def cpu_baseline(x, got_heads):
    """The oracle (scalar C restatement, kind "port") timed on this box's host cores on the SAME 4096 frames,
    3 repetitions (~20 s of CPU work); also the in-bench parity check of the GPU result."""
    from oracle.oracle import Oracle
    orc = Oracle()
    cores = min(len(os.sched_getaffinity(0)), 16)     # a 1-GPU box's CPU share is 16 cores
    reps, best = 3, None
    ref = None
    for _ in range(reps):
        t0 = time.perf_counter()
        ref = orc.run(x, threads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter()
    orc.run(x[:512], threads=1)
    one = 512 / (time.perf_counter() - t0)
    mism = int((ref != got_heads).sum())
    return dict(value=round(x.shape[0] / best, 1), unit="images/s", cores=cores, kind="port",
                sample=f"{reps} x {x.shape[0]} frames (the bench batch), best of {reps}, {cores} threads; "
                       f"single thread: {one:.0f} images/s",
                single_thread_images_per_s=round(one, 1)), mism


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (one step = one 4096-frame batch per GPU, ~0.23 ms)")
    ap.add_argument("--warmup", type=int, default=100, help="untimed steps; the first ~50 steps after idle run ~6%% slower (clock ramp)")
    ap.add_argument("--frames-per-wg", type=int, default=0)
    ap.add_argument("--waves-per-wg", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of the N>1 code path: ranks may share one GPU, collectives go through host copies)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    yf = importlib.import_module("stm32h7-yolo_amd")
    sharding = importlib.import_module("stm32h7-yolo_amd.sharding")
    net = yf.Network(device=dev_index, frames_per_wg=args.frames_per_wg, waves_per_wg=args.waves_per_wg).init()

    n, n_total = FRAMES_PER_GPU, FRAMES_PER_GPU * world
    a, b = sharding.shard_range(n_total, rank, world)
    assert b - a == n
    x = np.random.default_rng([1, rank]).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
    if rank == 0:       # golden frames inside the timed batch
        x[:6] = np.fromfile(os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)
    d_in = torch.from_numpy(x).to(dev)
    cap = 16
    # Per-rank output record: [heads n x 882 B | detection records n x cap x 28 B | counts n x 4 B] in ONE buffer, so that
    # the exchange at N > 1 is ONE all-gather per step.  Two such buffers alternate: the all-gather of step k runs on
    # RCCL's stream while the kernel of step k+1 fills the other buffer (collective overlapped with compute).
    off_d = (n * 882 + 15) & ~15
    off_c = off_d + n * cap * 28
    rec_bytes = (off_c + n * 4 + 15) & ~15
    n_buf = 2 if world > 1 else 1
    local = [torch.zeros((rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
    gath = [torch.zeros((world * rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(n_buf)] if world > 1 else []
    pending = [None] * n_buf
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def views(buf, r=0):
        base = buf[r * rec_bytes:(r + 1) * rec_bytes]
        return (base[:n * 882].view(torch.int8).view(n, 7, 7, 18), base[off_d:off_c].view(n, cap, 28),
                base[off_c:off_c + n * 4].view(torch.int32))

    def launch(i):
        p = local[i].data_ptr()
        # ONE launch per step: the fused network kernel also decodes the boxes of its frames (heads still in LDS)
        net.run_decode_device(d_in.data_ptr(), p, n, p + off_d, p + off_c, cap, yf.YF_DECODE_PY, 1.0, 1.0, sp)

    step_no = 0

    def step():
        nonlocal step_no
        i = step_no % n_buf
        step_no += 1
        if pending[i] is not None:          # the gather that last read this buffer must be done before it is overwritten
            pending[i].wait()
            pending[i] = None
        launch(i)
        if world > 1:       # every rank ends up with all heads and all detection records (RCCL over xGMI)
            if args.backend == "nccl":
                pending[i] = dist.all_gather_into_tensor(gath[i], local[i], async_op=True)
            else:           # rehearsal: the same exchange through host copies
                parts = [torch.empty((rec_bytes,), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, local[i].cpu())
                gath[i].copy_(torch.cat(parts))

    def drain():
        for i in range(n_buf):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    for _ in range(args.warmup):
        step()
    # One step is ONE kernel launch, so the fused kernel's average duration is the HIP-event time of the whole timed
    # region on the launch stream divided by the steps (a per-step event pair costs ~7 us of pipeline drain per step).
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_begin.record(stream)
    for k in range(args.steps):
        step()
    drain()
    ev_end.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(ev_begin.elapsed_time(ev_end)) / args.steps
    last = (step_no - 1) % n_buf
    if world > 1:           # the timed region above also holds the collectives: time the kernel alone, same stream, same inputs
        k_only = 100
        ev_begin.record(stream)
        for _ in range(k_only):
            launch(last)
        ev_end.record(stream)
        torch.cuda.synchronize()
        kernel_ms = float(ev_begin.elapsed_time(ev_end)) / k_only

    d_heads, d_dets, d_counts = views(local[last])
    heads = d_heads.cpu().numpy()
    ok_gather = True
    if world > 1:   # every rank must hold every rank's record (heads, detections, counts), in rank = frame order
        ok_gather = bool(torch.equal(gath[last][rank * rec_bytes:(rank + 1) * rec_bytes], local[last]))
        g_heads = torch.cat([views(gath[last], r)[0] for r in range(world)])        # [n_total, 7, 7, 18]
        ok_gather = ok_gather and tuple(g_heads.shape) == (n_total, 7, 7, 18) and bool(torch.equal(g_heads[a:b], d_heads))
        flag = torch.tensor([int(ok_gather)], device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_gather = bool(flag.item())

    if rank == 0:
        value = n_total * args.steps / elapsed
        line = {
            "metric": "images/sec int8 YOLO-face 56x56", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch=4096 int8 YOLO-face 56x56x3 frames per GPU, fused LDS-resident "
                                   "forward + GPU box decode" + (f" + RCCL all-gather of {n_total} heads and detection records" if world > 1 else ""),
                       "frames_per_gpu": n, "global_batch": n_total, "frame_bytes_in": 9408, "frame_bytes_out": 882,
                       "kernel": net.kernel_name, "parallelism": f"batch-shard x{world}, all-gather heads" if world > 1 else "single GPU"},
        }
        achieved = n * ALGO_BYTES_PER_FRAME / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # PMC passes are separate rocprofv3 runs (profiles/README.md)
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                            "kernel": net.kernel_name, "kernel_ms": round(kernel_ms, 4),
                            "algorithmic_bytes_per_launch": n * ALGO_BYTES_PER_FRAME}
        tops = n * DENSE_OPS_PER_FRAME / (kernel_ms * 1e-3) / 1e12
        line["roofline_mfma"] = {"bound": "mfma", "achieved": round(tops, 3), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                                 "frac": round(tops / MFMA_I8_PEAK_TOPS, 6)}
        if world == 1:
            cb, mism = cpu_baseline(x, heads)
            line["cpu_baseline"] = cb
            line["parity"] = "bit-exact vs oracle on %d/%d frames" % (n - (mism > 0) * 1, n) if mism == 0 else f"MISMATCH: {mism} head bytes differ"
            # PCIe-inclusive rate through the reference ABI (host buffers): reported, never `value`
            t1 = time.perf_counter()
            net.run(x)
            line["pcie_inclusive_images_per_s"] = round(n / (time.perf_counter() - t1), 1)
            if mism:
                print(json.dumps(line))
                raise SystemExit("GPU result differs from the oracle")
        else:
            line["all_gather_ok"] = ok_gather
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
