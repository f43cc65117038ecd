#!/usr/bin/env python3
"""images/s of the int8 YOLO-face 56x56 forward on N MI355X (BASELINE.json metric).

A step = one pass of the hot path over this rank's batch of 4096 synthetic frames already resident in HBM: ONE launch of the
fused kernel (forward + box decode) and, at N > 1, one RCCL all-gather of the DETECTIONS (fixed-capacity records + counts;
the int8 heads stay on their rank unless --gather-heads).  Weak scaling: every rank owns 4096 frames (BASELINE configs[1] at
N=1, configs[2] = 32768 frames at N=8).  The timed loop rotates through 8 distinct input batches (308 MB per rank, more than
the 256 MB Infinity Cache), so the input bytes really come from HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...          (no launcher: starts the N ranks itself as a child torch.distributed.run, see self_launch)
"""
import argparse
import hashlib
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
N_INPUT_BATCHES = 8                          # 8 x 38.5 MB > the 256 MB Infinity Cache
ALGO_BYTES_PER_FRAME = 9408 + 882            # input + head, everything else stays in LDS
DENSE_OPS_PER_FRAME = 2 * 813792             # int8 ops eligible for MFMA (dense convs), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0                   # dense int8 MFMA = 2x the ~2.5 PF bf16 rate (MI355X_MICROARCH.md, Matrix cores)
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9             # MI355X_MICROARCH.md chip-level parameters
VALU_CYCLES_PER_INSTR = 4.0                  # issue cost of a wave64 VALU instruction: 4.3-4.5 cycles for most of the epilogue's
                                             # ops, 2.3-2.7 for add/and/shift/mov (profiles/r01_e_valu_issue_rates.txt)
A160_BYTES_PER_FRAME = 160 * 160 * 3 + 20 * 20 * 18          # 84 000 B: algorithmic bytes of the 160x160 variant
FP16_BYTES_PER_FRAME = 56 * 56 * 3 * 2 + 7 * 7 * 18 * 4      # fp16 frame in, fp32 logits out
CAMERA_BYTES_PER_FRAME = 112 * 112 * 2 + 7 * 7 * 18          # RGB565 camera frame in, int8 head out (the prepared 56x56 frame never exists in HBM)
PROFILE_CAMERA = "profiles/r06_camera/summary.json"
PROFILE_160 = "profiles/r06_160/summary.json"                # rocprofv3 summaries of `bench.py --only-secondary ...` (tools/profile_secondary.sh),
PROFILE_FP16 = "profiles/r06_fp16/summary.json"              # stamped with the build id they were taken on
PROFILE_TIES_UP = "profiles/r06_u_pmc_current.json"          # tools/profile_pmc.sh r06u --requant-rounding ties_up: the second kernel set's counters


def kernel_source_hash():
    """Identity of the RUNNING kernel build: the id baked into the loaded library (sha256 over the device sources and the
    compiler flags they were built with, csrc/Makefile BUILD_ID) -- not a hash of whatever sources lie next to it."""
    return importlib.import_module("stm32h7-yolo_amd").load().yf_network_build_id().decode()


def profile_counters(kernel_name):
    """Counters of the fused kernel from the committed rocprofv3 PMC summary (profiles/pmc_current.json: separate --pmc passes,
    tools/profile_pmc.sh).  They are only reported when that profile was taken on THIS build (same kernel name, same source
    hash); otherwise null plus the reason -- a stale number is worse than none."""
    p = os.path.join(ROOT, "profiles", "pmc_current.json")
    if not os.path.exists(p):
        return None, "profiles/pmc_current.json is missing"
    d = json.load(open(p))
    if d.get("source_hash") != kernel_source_hash():
        return None, f"profile {d.get('profile')} was taken on kernel sources {d.get('source_hash')}, this build is {kernel_source_hash()}"
    if d.get("kernel") != kernel_name:
        return None, f"profile is of {d.get('kernel')}, running {kernel_name}"
    return d, None


def stamped_profile(rel_path, hash_of):
    """A committed rocprofv3 summary of a secondary configuration (profiles/...): returned only when it was taken on the running build."""
    p = os.path.join(ROOT, rel_path)
    if not os.path.exists(p):
        return None, f"{rel_path} is missing"
    d = json.load(open(p))
    h = hash_of(d)
    if h != kernel_source_hash():
        return None, f"{rel_path} was taken on build {h}, this build is {kernel_source_hash()}"
    return d, None


def _stamped_traffic(rel_path):
    """roofline.traffic (+ the trace's kernel time) of a side configuration from its stamped summary, or null with the reason"""
    prof, why = stamped_profile(rel_path, lambda d: d.get("total", {}).get("source_hash"))
    if not prof:
        return {"traffic": None, "traffic_missing": why}
    out = {"traffic": round(prof["total"]["hbm_bytes_per_batch"]), "traffic_source": f"{rel_path} (FETCH_SIZE / WRITE_SIZE passes, bytes per step)"}
    if prof["total"].get("timed_kernel_us_sum"):
        out["kernel_us_in_trace"] = round(prof["total"]["timed_kernel_us_sum"], 2)
    return out


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# --requant-rounding: library rounding (include/yf_network.h YF_ROUND_*) and the oracle variant that states it (oracle/yf_oracle.h YFO_RV_*)
ROUNDINGS = {"ref": (0, 0), "ties_up": (1, 1), "ties_up_all": (2, 2), "single": (3, 4), "ties_up+generic": (0x101, 1)}   # +generic: that rounding on the reference rounding's kernels


def cpu_quota_cores():
    """CPU time this process's cgroup may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable.  A one-GPU box of the pool
    shows all of the host's cores in the affinity mask and limits the TIME: threads beyond the quota only take turns."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(x, got_heads, variant=0):
    """The oracle (scalar C restatement, kind "port") timed on this box's host cores on the SAME 4096 frames -- about 2 s of CPU work in all;
    also the in-bench parity check of the GPU result.  Three figures, each the MEDIAN of 3 repetitions ("(i) single thread and
    (ii) all cores"), each with its thread count:
      value / cores                 16 threads: the CPU share of a ONE-GPU box of this pool (gpurun: "16 for one GPU"), what rounds 1-5 reported
      all_cores_images_per_s        one thread per core this process may run on (affinity_cores: 256 on the driver's box) -- north_star's "the GPU
                                    box's own cores (core count stated)".  On a one-GPU box of this pool it comes out BELOW the 16-thread figure (13.6 k
                                    against 18.9 k): the affinity mask shows the host's 256 cores, the cgroup grants the CPU time of a share of them
                                    (cpu_quota_cores, reported when readable), and 256 threads then take turns
      single_thread_images_per_s    one thread, on 512 of the frames
    with the CPU model and the box's core count beside them.  Returns (dict, mismatching head bytes, the oracle's heads)."""
    from oracle.oracle import Oracle
    orc = Oracle()
    affinity = len(os.sched_getaffinity(0))
    cores = min(affinity, 16)
    reps = 3

    def median_rate(frames, threads):
        times, out = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = orc.run(frames, threads=threads, variant=variant)
            times.append(time.perf_counter() - t0)
        return frames.shape[0] / sorted(times)[reps // 2], frames.shape[0] / min(times), out
    rate, best, ref = median_rate(x, cores)
    all_rate, all_best, _ = median_rate(x, affinity) if affinity != cores else (rate, best, None)
    one, _, _ = median_rate(x[:512], 1)
    mism = int((ref != got_heads).sum())
    return dict(value=round(rate, 1), unit="images/s", cores=cores, kind="port",
                sample=f"{reps} x {x.shape[0]} frames (the bench batch), median of {reps}, {cores} threads (a one-GPU box's CPU share) of {affinity} usable "
                       f"({os.cpu_count()} in the box, {cpu_model()}); best {best:.0f} images/s; all {affinity} usable cores: {all_rate:.0f} images/s; "
                       f"single thread: {one:.0f} images/s",
                best_images_per_s=round(best, 1), all_cores_images_per_s=round(all_rate, 1), all_cores_threads=affinity,
                all_cores_best_images_per_s=round(all_best, 1), affinity_cores=affinity, box_cores=os.cpu_count(), cpu_model=cpu_model(),
                cpu_quota_cores=cpu_quota_cores(),
                single_thread_images_per_s=round(one, 1)), mism, ref


def event_time_ms(stream, fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return float(e0.elapsed_time(e1)) / iters


def settle(fn, ms):
    """Untimed launches of `fn` for about `ms` milliseconds: the same clock settle as the headline region gets (the CPU baseline that runs
    between the headline and the secondary configurations leaves the GPU idle for ~20 s, and an idle GPU ramps its clock over ~10 ms)."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def secondary_160(net, dev, stream, settle_ms=60.0, iters=20):
    """BASELINE configs[4]: 160x160, batch 1024 (three banded launches per step)."""
    sp = stream.cuda_stream
    rng = np.random.default_rng(4)
    n = 1024
    d_in = torch.from_numpy(rng.integers(-128, 128, (n, 160, 160, 3), dtype=np.int8)).to(dev)
    d_out = torch.zeros((n, 20, 20, 18), dtype=torch.int8, device=dev)
    run = lambda: net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), n, sp)      # noqa: E731
    for _ in range(3):
        run()
    settle(run, settle_ms)
    ms = event_time_ms(stream, run, iters)
    gbs = n * A160_BYTES_PER_FRAME / (ms * 1e-3) / 1e9
    res = {"workload": "BASELINE configs[4]: batch=1024 int8 160x160x3 frames, one GPU", "ms_per_step": round(ms, 4), "timed_steps": iters,
           "images_per_s": round(n / ms * 1e3, 1), "algorithmic_bytes_per_step": n * A160_BYTES_PER_FRAME,
           "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5)},
           "kernel": "band_k1, band_k23, band_k4: three launches, each a group of fused stages over row bands staged through LDS (DESIGN.md, 160x160)",
           "kernel_source_hash": kernel_source_hash()}
    prof, why = stamped_profile(PROFILE_160, lambda d: d.get("total", {}).get("source_hash"))
    res["roofline"]["traffic"] = round(prof["total"]["hbm_bytes_per_batch"]) if prof else None
    res["roofline"]["traffic_source" if prof else "traffic_missing"] = \
        f"{PROFILE_160} (FETCH_SIZE / WRITE_SIZE passes, bytes per 1024-frame batch)" if prof else why
    if prof and prof["total"].get("timed_kernel_us_sum"):
        res["roofline"]["kernel_us_in_trace"] = round(prof["total"]["timed_kernel_us_sum"], 2)     # the three kernels of a step, timed launches of the profiled bench command
    return res


def secondary_camera(net, dev, stream, settle_ms=60.0, iters=20):
    """Camera-format pipeline: 112x112 RGB565 frames -> heads + firmware-mode boxes, the frame preparation fused
    into the kernel's input staging (ONE launch) against the two-launch form (separate preparation kernel)."""
    sp = stream.cuda_stream
    rng = np.random.default_rng(5)
    n = 4096
    d_raw = torch.from_numpy(rng.integers(0, 256, (n, 112 * 112 * 2), dtype=np.uint8)).to(dev)
    d_x = torch.zeros((n, 56, 56, 3), dtype=torch.int8, device=dev)
    d_h = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device=dev)
    d_d = torch.zeros((n, 4, 28), dtype=torch.uint8, device=dev)
    d_c = torch.zeros((n,), dtype=torch.int32, device=dev)
    one = lambda: net.run_camera_device(d_raw.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), 4, stream=sp)      # noqa: E731

    def two():
        net.prepare_rgb565_device(d_raw.data_ptr(), d_x.data_ptr(), n, sp)
        net.run_decode_device(d_x.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), 4, 1, 1.0, 1.0, sp)
    for _ in range(3):
        one(); two()
    settle(one, settle_ms)
    ms1, ms2 = event_time_ms(stream, one, iters), event_time_ms(stream, two, iters)
    gbs = n * CAMERA_BYTES_PER_FRAME / (ms1 * 1e-3) / 1e9
    return {"workload": "batch=4096 camera frames (112x112 big-endian RGB565, 25 088 B each) -> int8 heads + firmware-mode boxes",
            "ms_per_step": round(ms1, 4), "timed_steps": iters, "images_per_s": round(n / ms1 * 1e3, 1), "algorithmic_bytes_per_step": n * CAMERA_BYTES_PER_FRAME,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None,
                         **_stamped_traffic(PROFILE_CAMERA)},
            "kernel": "yoloface56_fused<F=2,NW=8,RGB565 input>", "kernel_source_hash": kernel_source_hash(),
            "two_launch_form_ms_per_step": round(ms2, 4), "two_launch_form_images_per_s": round(n / ms2 * 1e3, 1)}


def secondary_fp16(net, dev, stream, settle_ms=60.0, iters=20):
    """BASELINE configs[3]: fp16 weights of the reference's ONNX export, batch 4096, one fused launch per step."""
    sp = stream.cuda_stream
    rng = np.random.default_rng(6)
    n = 4096
    net.fp16_init()
    d_in = torch.from_numpy((rng.integers(0, 256, (n, 56, 56, 3)).astype(np.float32) / 255).astype(np.float16)).to(dev)
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device=dev)
    run = lambda: net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n, sp)              # noqa: E731
    for _ in range(3):
        run()
    settle(run, settle_ms)
    ms = event_time_ms(stream, run, iters)
    gbs = n * FP16_BYTES_PER_FRAME / (ms * 1e-3) / 1e9
    res = {"workload": "BASELINE configs[3]: batch=4096 fp16 56x56x3 frames (weights of the reference's ONNX export), one GPU",
           "ms_per_step": round(ms, 4), "timed_steps": iters, "images_per_s": round(n / ms * 1e3, 1), "algorithmic_bytes_per_step": n * FP16_BYTES_PER_FRAME,
           "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5)},
           "dtype": "f16 (f32 accumulate)", "kernel": "yoloface56_f16_fused<8>", "kernel_source_hash": kernel_source_hash()}
    prof, why = stamped_profile(PROFILE_FP16, lambda d: d.get("source_hash"))
    res["roofline"]["traffic"] = round(prof["hbm_bytes_per_launch"]) if prof else None
    res["roofline"]["traffic_source" if prof else "traffic_missing"] = \
        f"{PROFILE_FP16} (FETCH_SIZE / WRITE_SIZE passes, bytes per 4096-frame launch)" if prof else why
    if prof and prof.get("trace", {}).get("timed_avg_us"):
        res["roofline"]["kernel_us_in_trace"] = round(prof["trace"]["timed_avg_us"], 2)        # timed launches of the profiled bench command
    return res


def secondary_ties_up(net, dev, stream, settle_ms=60.0, iters=20):
    """The headline's workload under the OTHER likely definition of "tflite int8": rounding ties upward on the dense convs (what the default op resolver of the
    reference's script most probably computes; yf_network_set_requant_rounding).  Runs the second kernel set (three-instruction dense
    epilogue).  One stream, eight rotating 4096-frame batches in HBM, the fused decode in the launch -- like the headline's `pipelining.one_stream` figure; the
    first 256 heads are compared with the oracle's statement of that variant.  The network is back on the reference rounding afterwards."""
    import importlib
    yf = importlib.import_module("stm32h7-yolo_amd")
    sp = stream.cuda_stream
    n, nb, cap = 4096, 8, 4
    d_in = torch.randint(-128, 128, (nb, n, 56, 56, 3), dtype=torch.int8, device=dev)
    d_h = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device=dev)
    d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device=dev)
    d_c = torch.zeros((n,), dtype=torch.int32, device=dev)
    k = [0]

    def run():
        net.run_decode_device(d_in[k[0] % nb].data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, yf.YF_DECODE_PY, 1.0, 1.0, sp)
        k[0] += 1
    net.set_requant_rounding(yf.YF_ROUND_TIES_UP)
    try:
        kernel = net.kernel_name
        for _ in range(3):
            run()
        settle(run, settle_ms)
        ms = event_time_ms(stream, run, iters)
        k[0] = 0
        run()
        torch.cuda.synchronize()
        from oracle.oracle import Oracle, RV_UP_DENSE
        ok = bool(np.array_equal(d_h[:256].cpu().numpy(), Oracle().run(d_in[0, :256].cpu().numpy(), threads=min(len(os.sched_getaffinity(0)), 16), variant=RV_UP_DENSE)))
    finally:
        net.set_requant_rounding(yf.YF_ROUND_TFLITE_REF)
    gbs = n * ALGO_BYTES_PER_FRAME / (ms * 1e-3) / 1e9
    prof, why = stamped_profile(PROFILE_TIES_UP, lambda d: d.get("source_hash"))
    return {"workload": "BASELINE configs[1]'s batch (4096 int8 56x56x3 frames, 8 batches rotating in HBM) with requantisation rounding 'ties upward on the dense convs' "
                        "(yf_network_set_requant_rounding(YF_ROUND_TIES_UP)); one launch stream", "ms_per_step": round(ms, 4), "timed_steps": iters,
            "images_per_s": round(n / ms * 1e3, 1), "algorithmic_bytes_per_step": n * ALGO_BYTES_PER_FRAME,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                         **({"traffic": prof["hbm_bytes_per_launch"], "traffic_source": f"{PROFILE_TIES_UP} (FETCH_SIZE / WRITE_SIZE passes, bytes per launch)",
                             "kernel_us_in_trace": round(prof["kernel_trace_avg_ns"] / 1e3, 2), "valu_instructions_per_launch": prof.get("SQ_INSTS_VALU")}
                            if prof else {"traffic": None, "traffic_missing": why})},
            "kernel": kernel, "kernel_source_hash": kernel_source_hash(),
            "parity": "first 256 heads bit-exact vs the oracle's variant 'ties upward on the dense convs'" if ok else "MISMATCH vs the oracle's variant"}


SECONDARY = {"int8_160x160": secondary_160, "camera_rgb565_112x112": secondary_camera, "fp16_56x56": secondary_fp16, "int8_56x56_rounding_ties_up": secondary_ties_up}


def secondary_configs(net, dev, stream, settle_ms=60.0, iters=20, only=None):
    """The side configurations of BASELINE.json, timed after the headline region with HIP events on the launch stream, each after the
    headline's clock settle (`settle`).  Parity of all of them is the job of tests/test_gpu_parity.py; here only time.  `only` = one
    section's name: what `bench.py --only-secondary NAME` runs, which is the command the rocprofv3 summaries under profiles/r05_* are
    taken on -- the kernel time of a secondary line and its trace come from ONE command in ONE clock regime."""
    return {name: fn(net, dev, stream, settle_ms, iters) for name, fn in SECONDARY.items() if only in (None, name)}


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU (importing torch does not); it
    starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process (one rank per
    GPU, rendezvous on 127.0.0.1 at a free port), relays the child's output -- rank 0's single JSON line included -- and
    returns its exit code.  With WORLD_SIZE already set by a launcher this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (one step = one 4096-frame batch per GPU, ~0.2 ms)")
    ap.add_argument("--warmup", type=int, default=100, help="untimed steps; the first ~50 steps after idle run ~6%% slower (clock ramp)")
    ap.add_argument("--frames-per-wg", type=int, default=0)
    ap.add_argument("--waves-per-wg", type=int, default=0)
    ap.add_argument("--det-cap", type=int, default=4, help="detection records kept (and exchanged) per frame; the count is always the true count")
    ap.add_argument("--gather-heads", action="store_true", help="also all-gather the int8 heads (882 B per frame) at N > 1")
    ap.add_argument("--gather-every", type=int, default=1, metavar="K",
                    help="N > 1: ONE all-gather per K steps carrying K steps' records (default 1: the all-gather per step BASELINE.json names); the "
                         "exchange is latency-bound at 0.48 MB per rank and step, so K steps per collective cost one launch + handshake instead of K")
    ap.add_argument("--exchange-buffers", type=int, default=0, metavar="B",
                    help="N > 1: exchange buffers that alternate (0 = automatic).  A step's kernel waits for the gather that last read ITS buffer: with two "
                         "buffers that is the gather of two steps ago, which on a GPU filled by the kernels can only run once the step in between drains")
    ap.add_argument("--compact-records", action="store_true",
                    help="N > 1: send 12-byte wire records (cell + the firing anchor's six int8 head values; lossless, sharding.pack_compact) instead of the 28-byte yf_det")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2),
                    help="launch streams consecutive steps alternate between (default 2: the next step's workgroups start on CUs the previous launch "
                         "has drained, which hides the ~6.6 us of ramp / drain / dispatch gap a launch costs on one stream: tools/probe/batch_rate.py, two_stream.py)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 160x160 and fp16 side configurations")
    ap.add_argument("--only-secondary", choices=sorted(SECONDARY), default=None,
                    help="run ONLY this side configuration (after the same clock settle) and print its line: the command the rocprofv3 "
                         "summaries of the secondaries are taken on (tools/profile_secondary.sh), N = 1 only")
    ap.add_argument("--secondary-iters", type=int, default=20, help="timed steps of each side configuration")
    ap.add_argument("--clock-settle-ms", type=float, default=60.0,
                    help="before the W warm-up steps keep the GPU busy with untimed launches of the same kernel for this long: a GPU "
                         "that has idled runs its first ~10 ms at a lower engine clock (tools/step_probe.py), and W = 5 steps are 1 ms; 0 = off")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="N = 1 only: run the N > 1 code path -- process group (backend nccl = RCCL), per-step all-gather, gathered-record checks -- in a "
                         "ONE-rank group: the only way to execute the torch.distributed / RCCL calls of this file on a one-GPU box (RCCL refuses two ranks on one device)")
    ap.add_argument("--input-batches", type=int, default=N_INPUT_BATCHES, metavar="B",
                    help=f"distinct input batches the steps rotate through (default {N_INPUT_BATCHES}: 308 MB per rank, more than the 256 MB Infinity Cache, so a "
                         "launch reads its frames from HBM); a rehearsal with several ranks on ONE GPU lowers it so that the ranks do not generate 2.5 GB")
    ap.add_argument("--requant-rounding", choices=sorted(ROUNDINGS), default="ref",
                    help="which published rounding of TFLite's requantisation the network computes (yf_network_set_requant_rounding): ref = the builtin reference "
                         "kernels, the metric's definition (default); ties_up = dense convs as ruy rounds them (the default resolver of tflite_prediction.py:23); "
                         "same kernels, other constants -- every check of this run then compares with the oracle's statement of THAT variant")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of the N>1 code path: ranks may share one GPU, collectives go through host copies)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))      # plain `python bench.py --gpus N`: start the N ranks, relay their line
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if args.rccl_one_rank:
        if world != 1:
            raise SystemExit("--rccl-one-rank is an N = 1 rehearsal")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.update(RANK="0", WORLD_SIZE="1")
    dist_on = world > 1 or args.rccl_one_rank         # the exchange and its checks run (world = 1 with --rccl-one-rank: a one-rank group)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    yf = importlib.import_module("stm32h7-yolo_amd")
    sharding = importlib.import_module("stm32h7-yolo_amd.sharding")
    net = yf.Network(device=dev_index, frames_per_wg=args.frames_per_wg, waves_per_wg=args.waves_per_wg).init()
    rounding, variant = ROUNDINGS[args.requant_rounding]
    if rounding:
        net.set_requant_rounding(rounding)

    if args.only_secondary:
        if world != 1:
            raise SystemExit("--only-secondary is a one-GPU run")
        sec = secondary_configs(net, dev, torch.cuda.current_stream(), args.clock_settle_ms, args.secondary_iters, only=args.only_secondary)
        print(json.dumps({"only_secondary": args.only_secondary, "n_gpus": 1, "clock_settle_ms": args.clock_settle_ms, "data": "synthetic",
                          "secondary": sec}), flush=True)
        return

    n, n_total = FRAMES_PER_GPU, FRAMES_PER_GPU * world
    a, b = sharding.shard_range(n_total, rank, world)
    assert b - a == n
    golden_in = np.fromfile(os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)
    xs = []
    NB = max(1, args.input_batches)
    for k in range(NB):
        xk = np.random.default_rng([1, rank, k]).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
        if rank == 0 and k == 0:    # golden frames inside the timed batch
            xk[:6] = golden_in
        xs.append(xk)
    d_ins = [torch.from_numpy(xk).to(dev) for xk in xs]
    x = xs[0]
    cap = args.det_cap
    # Per-rank exchange record [detection records n x cap x 28 B | counts n x 4 B (| heads n x 882 B)] in ONE buffer: ONE all-gather
    # per step at N > 1.  Two buffers alternate: the all-gather of step k runs on RCCL's stream while the kernel of step k+1 fills
    # the other buffer (sharding.DetectionExchange holds the layout and the double-buffer protocol; the CPU gloo test runs it too).
    # Launch streams: consecutive steps alternate between args.streams streams (one stream: the process's current stream, as in rounds 1-4).  Batches are
    # independent, every step is still ONE launch over ONE batch; on two streams the workgroups of step k+1 start on CUs step k has left, so a launch's
    # ramp, drain and dispatch gap overlap the neighbour step instead of adding up (the kernel's OWN duration is what `roofline` reports, below).
    streams = [torch.cuda.current_stream()] if args.streams == 1 else [torch.cuda.Stream() for _ in range(args.streams)]
    S = len(streams)
    ex = sharding.DetectionExchange(n, cap, world, dev, gather_heads=args.gather_heads, backend=args.backend if dist_on else None, always=args.rccl_one_rank,
                                    gather_every=args.gather_every, compact=args.compact_records, launch_streams=streams if S > 1 else None, n_buf=args.exchange_buffers or None,
                                    packer=lambda d, c, h, w, nn, cp: net.pack_detections_device(d, c, h, w, nn, cp, torch.cuda.current_stream().cuda_stream),
                                    unpacker=lambda w, c, h, nn, cp: net.unpack_detections_device(w, c, h, nn, cp, torch.cuda.current_stream().cuda_stream))
    rec_bytes = ex.wire_rec_bytes
    stream = streams[0]
    Slot0 = sharding.Slot(0, 0)

    def launch(slot, k=0, s=stream):
        # ONE launch per step: the fused network kernel also decodes the boxes of its frames (heads still in LDS)
        net.run_decode_device(d_ins[k].data_ptr(), ex.heads(slot).data_ptr(), n, ex.dets_ptr(slot), ex.counts_ptr(slot), cap, yf.YF_DECODE_PY, 1.0, 1.0, s.cuda_stream)

    step_no = 0

    def step(batch=None, stream_no=None):
        """one step; `batch` / `stream_no` override the rotation (the check steps behind the timed region, the one-stream steps)"""
        nonlocal step_no
        k, s = (step_no % NB if batch is None else batch), streams[(step_no if stream_no is None else stream_no) % S]
        step_no += 1
        if not dist_on:                     # nothing to order against: no collective reads the buffers
            slot = ex.acquire()
            launch(slot, k, s)
            return slot
        with torch.cuda.stream(s):          # the exchange orders itself against the CURRENT stream: make it the step's launch stream
            slot = ex.acquire()             # waits for the gather that last read the slot's buffer
            launch(slot, k, s)
            ex.exchange(slot)               # every rank ends up with every rank's detection records and counts (RCCL over xGMI); one collective per K steps
        return slot

    drain = ex.drain

    # Clock settle (untimed, reported as config.clock_settle_ms): the engine clock of an idle GPU ramps up over the first ~10 ms
    # of work -- the first ~50 launches run ~6 % slower -- and a short warm-up (W = 5 steps = 1 ms) would leave the timed
    # region inside that ramp.  Same kernel, same buffers, no collectives; the W warm-up steps and the K timed steps follow.
    # The settle launches rotate through the input batches like the timed steps, so every full-batch launch of a run reads its frames from HBM and the
    # kernel-trace average of the whole command (profiles/) is the average of launches like the timed ones (one batch alone stays in the Infinity Cache
    # and runs ~3 % faster: round 4's first trace averaged 140.9 us over launches whose timed part took 145.6 us).
    settled_ms, settle_no = 0.0, 0

    def settle_launch():
        nonlocal settle_no
        launch(Slot0, settle_no % NB)
        settle_no += 1

    while settled_ms < args.clock_settle_ms:
        settled_ms += event_time_ms(stream, settle_launch, 25) * 25
    for _ in range(args.warmup):
        step()
    # One step is ONE kernel launch, so the fused kernel's average duration is the HIP-event time of the whole timed
    # region on the launch stream divided by the steps (a per-step event pair costs ~7 us of pipeline drain per step).
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_begin.record(stream)
    for _ in range(args.steps):
        step()
    drain()
    for s_other in streams[1:]:
        stream.wait_stream(s_other)         # the closing event covers the launches of every stream
    ev_end.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0          # this rank's K steps; the MAX over ranks below is the job's time (a rank that finishes early waits in the barrier,
    if dist_on:                               # and the barrier's own collective -- ~0.1 ms against a 2.9 ms region at the driver's flags -- is not a step)
        dist.barrier()
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    region_ms = float(ev_begin.elapsed_time(ev_end)) / args.steps
    kernel_ms = region_ms
    if dist_on or S > 1:  # the timed region above also holds the collectives / overlaps consecutive launches: time the kernel ALONE -- one stream, same
        kernel_ms = event_time_ms(stream, settle_launch, 100)      # inputs, back to back; with one stream and one rank the timed region is that already

    # ---- the same K steps on ONE stream, between the same kind of brackets (N = 1 line only): what rounds 1-4 reported as `value`, so that a reader of
    # the line can separate launch policy (two streams) from kernel progress without profiles/
    one_stream = None
    if S > 1 and not dist_on:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(stream_no=0)
        torch.cuda.synchronize()
        one_stream = (time.perf_counter() - t1) / args.steps

    # ---- correctness of what was just timed: one more step on batch 0 (the one holding the golden frames) PER LAUNCH STREAM AND EXCHANGE BUFFER -- the
    # timed region ran on all of them (round 5 checked stream 0 / buffer 0 only) --, every one compared with the oracle below
    drain()
    step_no = 0
    ex.step_no = 0
    n_check_steps = max(S, ex.n_buf) * ex.K
    check_slots = [step(batch=0) for _ in range(n_check_steps)]
    drain()
    torch.cuda.synchronize()
    d_dets, d_counts = ex.views(ex.local[0])
    heads = ex.heads(Slot0).view(torch.int8).view(n, 7, 7, 18).cpu().numpy()
    dets = d_dets.cpu().numpy().view(yf.DET_DTYPE).reshape(n, cap)
    counts = d_counts.cpu().numpy()
    from oracle.oracle import Oracle
    orc = Oracle()
    n_chk = n if not dist_on else 256            # at N > 1 every rank checks a slice of ITS shard (the full check is the N = 1 run's)
    problems = []
    for j, sl in enumerate(check_slots[1:], 1):  # every other (stream, buffer, slot) pair produced what (stream 0, buffer 0, slot 0) produced, bit for bit: heads, records, counts
        r_j, c_j = ex.views(ex.local[sl.i], 0, sl.k)
        kept_j = torch.arange(cap, device=dev)[None, :] < d_counts.clamp(max=cap)[:, None]     # (record slots beyond a frame's count are never written: stale bytes)
        if not (torch.equal(ex.heads(sl), ex.heads(Slot0)) and torch.equal(c_j, d_counts) and torch.equal(r_j[kept_j], d_dets[kept_j])):
            problems.append(f"rank {rank}: check step {j} (stream {j % S}, buffer {sl.i}, slot {sl.k}) differs from check step 0")
    ref_heads = orc.run(x[:n_chk], threads=min(len(os.sched_getaffinity(0)), 16), variant=variant) if dist_on else None
    if dist_on and not np.array_equal(heads[:n_chk], ref_heads):
        problems.append(f"rank {rank}: heads differ from the oracle on its first {n_chk} frames")
    if dist_on and os.environ.get("YF_BENCH_TEST_FAIL_RANK") == str(rank):         # read by ONE test only: rehearses "a rank's check fails -> every
        problems.append(f"rank {rank}: parity failure forced by YF_BENCH_TEST_FAIL_RANK")   # rank exits non-zero after rank 0 has printed its line"
    for f in range(min(n_chk, 64)):             # decoded records of the first frames against the oracle's decode of the GPU heads
        want = orc.decode_py(heads[f], f, 1.0, 1.0)
        got = [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["q_conf"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in dets[f, :min(cap, counts[f])]]
        if counts[f] != len(want) or got != [(w[1], w[2], w[3], w[4], w[6], w[7], w[8], w[9]) for w in want][:cap]:
            problems.append(f"rank {rank}: detection records of frame {f} differ from the oracle's decode")
            break
    if rank == 0 and variant == 0:              # the golden frames' detections are committed (tests/golden/golden_meta.json; reference rounding)
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_meta.json")))
        for f, fr in enumerate(meta["frames"]):
            want = [(g["anchor"], g["row"], g["col"], g["x1"], g["y1"], g["x2"], g["y2"]) for g in fr["detections_py"]]
            got = [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in dets[f, :min(cap, counts[f])]]
            if counts[f] != len(want) or got != want[:cap]:
                problems.append(f"golden frame {f}: detections differ from tests/golden/golden_meta.json")
    ok_gather, ranks_sampled = True, []
    if dist_on:   # every rank must hold every rank's records, in rank = frame order
        if args.compact_records:    # the wire carries cells + logits: decode the sparse heads they stand for with the library's own decode -> the sender's records
            sparse = ex.gathered_sparse_heads(Slot0).contiguous()
            r_dets = torch.zeros((n_total, cap, 28), dtype=torch.uint8, device=dev)
            r_counts = torch.zeros((n_total,), dtype=torch.int32, device=dev)
            net.decode_device(sparse.data_ptr(), n_total, r_dets.data_ptr(), r_counts.data_ptr(), cap, yf.YF_DECODE_PY, 1.0, 1.0)
            torch.cuda.synchronize()
            kept = torch.arange(cap, device=dev)[None, :] < d_counts.clamp(max=cap)[:, None]
            ok_gather = ok_gather and bool(torch.equal(r_counts[a:b], d_counts.clamp(max=cap)))
            ok_gather = ok_gather and bool(torch.equal(r_dets[a:b][kept][:, 4:], d_dets[kept][:, 4:]))        # every field but the (positional) frame index
            g_frames = r_dets[:, 0, :4].contiguous().view(torch.int32).view(-1).cpu().numpy()
            frame_of = lambda g: g_frames[g] % n                                       # noqa: E731  (decoded over n_total frames: the index is the global position)
            ok_gather = ok_gather and all(int(g_frames[g]) == g for g in ex.rank_major_samples(Slot0))
        else:
            g_frames = ex.gathered_records(Slot0)[:, 0, :4].contiguous().view(torch.int32).view(-1).cpu().numpy()
            frame_of = lambda g: g_frames[g]                                           # noqa: E731  (record k of rank r carries its LOCAL frame index)
        # this rank's block at its place, all ranks' counts as one array, and the first firing frames of EVERY rank's shard in rank-major order
        ok_rm, ranks_sampled = ex.gathered_is_rank_major(Slot0, rank, d_counts, frame_of)
        ok_gather = ok_gather and ok_rm and len(ranks_sampled) == world
        flag = torch.tensor([int(ok_gather and not problems)], device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = bool(flag.item())
    else:
        all_ok = not problems

    fail = None
    if rank == 0:
        value = n_total * args.steps / elapsed
        exch = (f" + RCCL all-gather of detection records (cap {cap}{', 12-byte wire form' if args.compact_records else ''}) and counts" + (" and heads" if args.gather_heads else "")
                + (f", one collective per {args.gather_every} steps" if args.gather_every > 1 else "")) if dist_on else ""
        line = {
            "metric": "images/sec int8 YOLO-face 56x56", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch=4096 int8 YOLO-face 56x56x3 frames per GPU, fused LDS-resident "
                                   "forward + GPU box decode" + exch,
                       "frames_per_gpu": n, "global_batch": n_total, "frame_bytes_in": 9408, "frame_bytes_out": 882,
                       "clock_settle_ms": round(settled_ms, 1), "input_batches_rotated": NB, "input_bytes_resident": NB * n * 9408,
                       "check_steps": n_check_steps,
                       "kernel": net.kernel_name, "kernel_source_hash": kernel_source_hash(), "requant_rounding": args.requant_rounding,
                       "parallelism": f"batch-shard x{world}, all-gather of detections" if dist_on else "single GPU",
                       "exchange_bytes_per_rank_per_step": rec_bytes if dist_on else 0, "gather_every": args.gather_every,
                       "collectives_issued": ex.collectives, "launch_streams": S,
                       **({"rehearsal": "--rccl-one-rank: the N > 1 code path in a ONE-rank RCCL group (not a multi-GPU measurement)"} if args.rccl_one_rank else {})},
        }
        achieved = n * ALGO_BYTES_PER_FRAME / (kernel_ms * 1e-3) / 1e9
        prof, why = profile_counters(net.kernel_name)
        line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": prof["hbm_bytes_per_launch"] if prof else None,
                            "traffic_source": (f"{prof['profile']} (kernel sources {prof['source_hash']})" if prof else why),
                            "kernel": net.kernel_name, "kernel_ms": round(kernel_ms, 4),
                            "kernel_ms_is": ("the timed region's HIP-event time / steps (one stream, one launch per step, back to back)" if (S == 1 and not dist_on) else
                                             "the kernel's OWN average duration: HIP events around 100 back-to-back launches on ONE stream after the timed region "
                                             "(same inputs); the rocprofv3 trace of the one-stream launches of this command agrees (profiles/)"),
                            "algorithmic_bytes_per_launch": n * ALGO_BYTES_PER_FRAME}
        if S > 1:   # the JOB's rate against the KERNEL's duration: consecutive steps overlap, so a step takes less than a kernel lasts -- by design, not by mistake
            line["pipelining"] = {"launch_streams": S, "timed_region_ms_per_step": round(region_ms, 4), "kernel_ms_alone": round(kernel_ms, 4),
                                  "hidden_per_step_us": round((kernel_ms - region_ms) * 1e3, 2),
                                  **({"one_stream_ms_per_step": round(one_stream * 1e3, 4), "one_stream_images_per_s": round(n_total / one_stream, 1),
                                      "one_stream_is": f"the same {args.steps} steps issued to ONE stream right after the timed region (wall clock between two device "
                                                       "synchronisations): the figure rounds 1-4 reported as `value`; value / this = what the second launch stream adds"}
                                     if one_stream else {}),
                                  "note": "steps alternate between two HIP streams: the workgroups of step k+1 start on CUs step k has drained, so a launch's ramp, drain and "
                                          "dispatch gap (6.6 us of fixed cost per launch on one stream, tools/probe/batch_rate.py) overlap the neighbour step.  "
                                          "ms_per_step < roofline.kernel_ms follows from that; `roofline` prices the kernel alone (--streams 1 reproduces rounds 1-4)"}
        tops = n * DENSE_OPS_PER_FRAME / (kernel_ms * 1e-3) / 1e12
        line["roofline_mfma"] = {"bound": "mfma", "achieved": round(tops, 3), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                                 "frac": round(tops / MFMA_I8_PEAK_TOPS, 6)}
        # The bound that binds in practice: VALU issue (requantisation epilogue, index arithmetic, pooling).  Instructions per
        # launch come from the PMC profile of this build; peak = every SIMD issuing one wave64 VALU instruction per 4 cycles.
        valu_peak = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INSTR
        if prof and prof.get("SQ_INSTS_VALU"):
            ips = prof["SQ_INSTS_VALU"] / (kernel_ms * 1e-3)
            line["roofline_valu"] = {"bound": "valu-issue", "achieved": round(ips / 1e9, 2), "peak": round(valu_peak / 1e9, 1), "unit": "G wave-instr/s",
                                     "frac": round(ips / valu_peak, 4), "valu_instructions_per_launch": prof["SQ_INSTS_VALU"],
                                     "valu_instructions_per_frame": round(prof["SQ_INSTS_VALU"] / n, 1), "cycles_per_instruction_assumed": VALU_CYCLES_PER_INSTR,
                                     "source": f"{prof['profile']} (kernel sources {prof['source_hash']})"}
        else:
            line["roofline_valu"] = {"bound": "valu-issue", "achieved": None, "peak": round(valu_peak / 1e9, 1), "unit": "G wave-instr/s", "frac": None, "source": why}
        # ... and the LDS pipe, which co-limits with the VALU port (DESIGN.md "What binds"): LDS-array cycles of the profile (SQ_LDS_IDX_ACTIVE,
        # summed over the CUs) against one LDS pipe per CU for the kernel's duration
        if prof and prof.get("SQ_LDS_IDX_ACTIVE"):
            cu_cycles = (SIMDS // 4) * CLOCK_HZ * (kernel_ms * 1e-3)
            line["roofline_lds"] = {"bound": "lds-pipe", "achieved": prof["SQ_LDS_IDX_ACTIVE"], "peak": round(cu_cycles), "unit": "LDS-array cycles per launch",
                                    "frac": round(prof["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4), "bank_conflict_cycles": prof.get("SQ_LDS_BANK_CONFLICT"),
                                    "lds_instructions_per_launch": prof.get("SQ_INSTS_LDS"), "source": f"{prof['profile']} (kernel sources {prof['source_hash']})"}
        if not dist_on:
            cb, mism, _ = cpu_baseline(x, heads, variant)
            line["cpu_baseline"] = cb
            line["parity"] = (("bit-exact vs oracle on %d/%d frames" % (n, n)) + ("; decoded boxes of the golden frames equal tests/golden" if variant == 0 else
                              f" (oracle variant of --requant-rounding {args.requant_rounding})")) if mism == 0 and not problems \
                else f"MISMATCH: {mism} head bytes differ; {problems}"
            # PCIe-inclusive rate through the reference ABI (host buffers): reported, never `value`
            # (the first call sizes the engine's device staging and starts its download thread: untimed; best of five after it)
            host_heads = net.run(x)
            if not np.array_equal(host_heads, heads):
                problems.append("ai_network_run on host buffers differs from the device path")
            best = float("inf")
            for _ in range(5):          # the caller's own arrays both ways, as the firmware's static in_data / out_data
                t1 = time.perf_counter()
                net.run(x, out=host_heads)
                best = min(best, time.perf_counter() - t1)
            line["pcie_inclusive_images_per_s"] = round(n / best, 1)
            if not args.no_secondary:   # the ABI's largest batch (ai_network_run takes a 16-bit n_batches): 65 535 frames, 617 MB of host frames
                xl = np.tile(x, (16, 1, 1, 1))[:65535]
                out_l = net.run(xl)
                best_l = float("inf")
                for _ in range(3):          # best of three (one call is 12-18 ms: a single sample caught a 3.7 M outlier among 5.3-5.7 M)
                    t1 = time.perf_counter()
                    net.run(xl, out=out_l)
                    best_l = min(best_l, time.perf_counter() - t1)
                line["pcie_inclusive_images_per_s_n65535"] = round(65535 / best_l, 1)
                if not np.array_equal(out_l[:n], heads) or not np.array_equal(out_l[-(65535 - 15 * n):], heads[:65535 - 15 * n]):
                    problems.append("ai_network_run on 65535 host frames differs from the device path")
                del xl, out_l
            # what the reference's own caller does (yoloface.c aiRun: n_batches = 1, host buffers): one frame per ai_network_run
            # (skipped with --no-secondary, which the profiling script uses: these 210 one-frame launches carry the headline kernel's
            # name and would be averaged into a rocprofv3 --stats summary of the command)
            if not args.no_secondary:
                lat = []
                for k in range(210):
                    t1 = time.perf_counter()
                    net.run(x[k:k + 1])
                    lat.append(time.perf_counter() - t1)
                line["single_frame_ai_network_run_us"] = round(float(np.median(lat[10:])) * 1e6, 1)
            if mism or problems:
                fail = f"GPU result differs from the oracle ({mism} head bytes; {problems})"
            elif not args.no_secondary:
                line["secondary"] = secondary_configs(net, dev, stream, args.clock_settle_ms, args.secondary_iters)
                if any(str(v.get("parity", "")).startswith("MISMATCH") for v in line["secondary"].values()):
                    fail = "a secondary configuration differs from the oracle"
        else:
            line["all_gather_ok"] = ok_gather
            line["rank_major_check"] = {"ranks_sampled": ranks_sampled, "note": "rank 0's view: firing frames sampled from every rank's shard carry that rank's local frame indices"}
            line["parity"] = "every rank: heads of its first 256 frames and decoded records of its first 64 frames equal the oracle; golden detections equal tests/golden" \
                if all_ok else f"FAILED on at least one rank (rank 0: {problems})"
            if not all_ok:
                fail = "all-gather or per-rank parity check failed"
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
        if not all_ok:
            raise SystemExit(fail or f"rank {rank}: check failed: {problems}")
    if fail:
        raise SystemExit(fail)


if __name__ == "__main__":
    main()
