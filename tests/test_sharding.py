"""Multi-GPU plumbing on CPU: world_size-2 gloo processes shard a batch, fill the packed detection records of their LOCAL
frames (the oracle stands in for the GPU engine here -- tests may use it as the checker) and exchange them through
sharding.DetectionExchange, the double-buffered all-gather bench.py runs at N > 1."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_range_partitions_the_batch():
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    for n in (0, 1, 2, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert sh.shard_range(32768, 3, 8) == (3 * 4096, 4 * 4096)      # BASELINE config 3: 4096 frames per rank


DET = np.dtype([("frame", "<i4"), ("anchor", "u1"), ("row", "u1"), ("col", "u1"), ("q_conf", "i1"),
                ("conf", "<f4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4")])      # yf_det, 28 bytes


def _records(orc, heads, cap):
    """What the fused kernel leaves in a rank's exchange buffer for these heads: cap records per frame (LOCAL frame index)
    and the true counts -- here produced by the oracle's decode (tests may use it as the checker)."""
    n = heads.shape[0]
    recs = np.zeros((n, cap), DET)
    counts = np.zeros((n,), np.int32)
    for f in range(n):
        d = orc.decode_py(heads[f], f, 1.0, 1.0)
        counts[f] = len(d)
        for k, r in enumerate(d[:cap]):
            recs[f, k] = r
    return recs, counts


def _worker(rank, world, port, n_local, steps, gather_heads, gather_every, compact, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    from oracle.oracle import Oracle
    orc = Oracle()
    cap, K = 4, gather_every
    ex = sh.DetectionExchange(n_local, cap, world, "cpu", gather_heads=gather_heads, gather_every=K, compact=compact)
    assert ex.n_buf == 2 and ex.rec_bytes % 16 == 0 and ex.off_c % 16 == 0 and ex.wire_rec_bytes % 16 == 0
    assert ex.wire_rec_bytes < ex.rec_bytes if compact else ex.wire_rec_bytes == ex.rec_bytes
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)   # real images: they fire
    ok = True

    def pick_of(step):     # a step's global batch: world * n_local frames, rank r owns [r * n_local, (r + 1) * n_local)
        return (np.arange(world * n_local) * 5 + 7 * step) % frames.shape[0]

    def gathered_is(slot, step):
        """every rank's records / counts (/ heads) of `step`, found in `slot` of the gathered buffer, in rank = frame order"""
        full_heads = orc.run(frames[pick_of(step)])
        want_r, want_c = zip(*[_records(orc, full_heads[r * n_local:(r + 1) * n_local], cap) for r in range(world)])
        good = np.array_equal(ex.gathered_counts(slot).numpy(), np.concatenate(want_c))
        if compact:     # 12-byte wire records: the sparse heads they stand for, decoded (by the checker), are the senders' records
            sparse = ex.gathered_sparse_heads(slot).numpy()
            for g in range(world * n_local):
                want = np.concatenate(want_r)[g][:min(cap, int(np.concatenate(want_c)[g]))]
                got = orc.decode_py(sparse[g], g % n_local, 1.0, 1.0)
                good = good and len(got) == len(want) and all(tuple(r)[1:] == tuple(w)[1:] for r, w in zip(np.array(got, DET) if got else [], want))
        else:
            good = good and np.array_equal(ex.gathered_records(slot).numpy().reshape(-1), np.concatenate(want_r).view(np.uint8).reshape(-1))
        if gather_heads:
            off = ex.w_off_h if compact else ex.off_h
            for r in range(world):
                base = (r * K + slot.k) * ex.wire_rec_bytes + off
                g = ex.gathered[slot.i][base: base + n_local * 882].numpy().view(np.int8)
                good = good and np.array_equal(g, full_heads[r * n_local:(r + 1) * n_local].reshape(-1))
        return bool(good)

    slots = []
    for step in range(steps):
        a, b = sh.shard_range(world * n_local, rank, world)
        heads = orc.run(frames[pick_of(step)[a:b]])
        recs, counts = _records(orc, heads, cap)
        slot = ex.acquire()                    # a buffer's first slot waits for the gather that last read the buffer (2 K steps earlier)
        slots.append(slot)
        assert (slot.i, slot.k) == ((step // K) % 2, step % K)
        if slot.k == 0:
            assert ex.pending[slot.i] is None
            if step >= 2 * K:
                ok = ok and ex.waits == step // K - 1
                for k in range(K):             # what that gather delivered: the K steps that filled this buffer last time
                    ok = ok and gathered_is(sh.Slot(slot.i, k), step - 2 * K + k)
        r_view, c_view = ex.views(ex.local[slot.i], 0, slot.k)
        r_view.copy_(torch.from_numpy(recs.view(np.uint8).reshape(n_local, cap, 28)))
        c_view.copy_(torch.from_numpy(counts))
        ex.heads(slot).copy_(torch.from_numpy(heads.reshape(-1).view(np.uint8)))
        ex.exchange(slot)
        assert (ex.pending[slot.i] is not None) == (slot.k == K - 1)        # ONE collective per K steps
    assert ex.collectives == steps // K
    ex.drain()                                 # sends a partly filled buffer too
    assert all(p is None for p in ex.pending) and ex.collectives == -(-steps // K)
    for step in range(max(0, (steps - 1) // K * K - K), steps):           # the steps of the last two buffers
        ok = ok and gathered_is(slots[step], step)
    ok = ok and ex.check_gathered(slots[-1], rank)
    # bench.py's rank = frame order check (sharding.gathered_is_rank_major): this rank's block at its place, every rank's counts as one array, and the first
    # firing frames of EVERY rank's shard carrying that rank's LOCAL frame indices -- the sample must reach into every shard (at N = 8 the first 512 firing
    # frames of the whole array, round 5's sample, all lie in rank 0's)
    last = slots[-1]
    if compact:
        sparse = ex.gathered_sparse_heads(last).numpy()
        frame_of = lambda g: orc.decode_py(sparse[g], g % n_local, 1.0, 1.0)[0][0]      # noqa: E731
    else:
        g_frames = ex.gathered_records(last)[:, 0, :4].contiguous().view(torch.int32).view(-1).numpy()
        frame_of = lambda g: g_frames[g]                                                # noqa: E731
    ok_rm, ranks_sampled = ex.gathered_is_rank_major(last, rank, ex.views(ex.local[last.i], 0, last.k)[1], frame_of, per_rank=2)
    ok = ok and ok_rm and ranks_sampled == list(range(world))
    if not compact:                            # a block at the wrong place IS caught: swap the blocks of the last two ranks in the gathered buffer
        g = ex.gathered[last.i]
        rb = ex.K * ex.wire_rec_bytes
        tmp = g[(world - 1) * rb:world * rb].clone()
        g[(world - 1) * rb:world * rb] = g[(world - 2) * rb:(world - 1) * rb]
        g[(world - 2) * rb:(world - 1) * rb] = tmp
        g_frames = ex.gathered_records(last)[:, 0, :4].contiguous().view(torch.int32).view(-1).numpy()
        caught = not ex.gathered_is_rank_major(last, rank, ex.views(ex.local[last.i], 0, last.k)[1], lambda gi: g_frames[gi], per_rank=2)[0]
        ok = ok and (caught or rank < world - 2)     # the two ranks whose own block moved see it; the others cannot (records carry LOCAL frame indices, a foreign
                                                     # shard looks the same wherever it lies) -- which is why EVERY rank checks its own block's place and bench.py
                                                     # MIN-reduces the verdicts
    #  a run that ended INSIDE a buffer and keeps stepping: drain() sent the partly filled buffer, the next acquire() must start a fresh
    # buffer at slot 0 -- not hand out slot k != 0 of the sent one, whose next send would retransmit the stale slots in front of it
    if K > 1 and steps % K:
        slot = ex.acquire()
        ok = ok and slot.k == 0 and ex.step_no % K == 1
        a, b = sh.shard_range(world * n_local, rank, world)
        heads = orc.run(frames[pick_of(steps)[a:b]])
        recs, counts = _records(orc, heads, cap)
        r_view, c_view = ex.views(ex.local[slot.i], 0, 0)
        r_view.copy_(torch.from_numpy(recs.view(np.uint8).reshape(n_local, cap, 28)))
        c_view.copy_(torch.from_numpy(counts))
        ex.heads(slot).copy_(torch.from_numpy(heads.reshape(-1).view(np.uint8)))
        ex.exchange(slot)
        ex.drain()
        ok = ok and gathered_is(slot, steps) and ex.filled == [0] * ex.n_buf
    if rank == 0:
        q.put((bool(ok), int(ex.gathered_counts(slots[-1]).sum()), ex.wire_rec_bytes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_local,steps,gather_heads,gather_every,compact",
                         [(4, 5, False, 1, False), (1, 3, True, 1, False), (7, 4, False, 1, False), (4, 7, False, 3, False), (4, 5, False, 1, True), (3, 6, True, 2, True)])
def test_two_rank_gloo_detection_exchange(n_local, steps, gather_heads, gather_every, compact):
    """World size 2 on CPU (gloo): the packed [records | counts (| heads)] exchange of bench.py with its two alternating
    buffers and asynchronous gathers -- sharding.DetectionExchange, the code the N-GPU run executes with RCCL -- must leave
    every rank's records, counts (and heads) on every rank in rank = frame order, step after step; also with ONE collective per
    K steps (--gather-every: a run that ends inside a buffer sends it from drain()) and with the 12-byte wire records
    (--compact-records: decoding the sparse heads they stand for gives the senders' records)."""
    port = 29500 + (os.getpid() + 17 * n_local + 131 * gather_every + 977 * compact) % 2000
    _run_ranks(2, port, n_local, steps, gather_heads, gather_every, compact)


def _run_ranks(world, port, n_local, steps, gather_heads, gather_every, compact):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, steps, gather_heads, gather_every, compact, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        ok, n_dets, rec_bytes = q.get(timeout=300)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ok
    assert n_dets > 0, "no frame fired: the record comparison would be vacuous"
    assert rec_bytes >= n_local * (4 * (12 if compact else 28) + 4)


@pytest.mark.parametrize("n_local,steps,gather_heads,gather_every,compact", [(3, 5, False, 1, False), (3, 5, True, 2, True)])
def test_eight_rank_gloo_detection_exchange(n_local, steps, gather_heads, gather_every, compact):
    """World size EIGHT on CPU (gloo) -- BASELINE configs[2]'s rank count; no round before this one ran anything above world size 2.  The same worker as the
    two-rank test: every rank's records / counts / heads on every rank in rank = frame order step after step (`gathered_is` walks all eight shards), the
    gathered-buffer indexing r * K + k at r up to 7, and bench.py's own rank-major check with its sample drawn from every rank's shard."""
    port = 31600 + (os.getpid() + 17 * n_local + 131 * gather_every + 977 * compact) % 2000
    _run_ranks(8, port, n_local, steps, gather_heads, gather_every, compact)


def test_compact_wire_records_round_trip():
    """pack_compact / unpack_compact on their own (CPU tensors): 12 bytes per record -- cell + the firing anchor's six int8 head values --, slots
    beyond a frame's count zeroed whatever stale bytes the record buffer holds, and the sparse heads rebuilt from the wire equal the real heads at
    every transmitted cell and are -128 everywhere else."""
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    from oracle.oracle import Oracle
    orc = Oracle()
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    heads = orc.run(frames)
    n, cap = heads.shape[0], 2                                  # cap 2: some frames have more candidates than slots
    recs, counts = _records(orc, heads, cap)
    assert (counts > cap).any() and (counts == 0).any()
    raw = recs.view(np.uint8).reshape(n, cap, 28).copy()
    raw[counts == 0] = 0xAB                                     # stale bytes where nothing fired: must not reach the wire
    wire = sh.pack_compact(torch.from_numpy(raw).view(-1), torch.from_numpy(counts), torch.from_numpy(heads.reshape(-1).view(np.uint8)), n, cap)
    w = wire.view(n, cap, 12).numpy()
    for f in range(n):
        for k in range(cap):
            if k < min(cap, counts[f]):
                a, r, c = int(recs[f, k]["anchor"]), int(recs[f, k]["row"]), int(recs[f, k]["col"])
                assert tuple(w[f, k, :4]) == (a, r, c, 0) and np.array_equal(w[f, k, 4:10].view(np.int8), heads[f, r, c, 6 * a:6 * a + 6]) and not w[f, k, 10:].any()
            else:
                assert not w[f, k].any()
    sparse = sh.unpack_compact(wire, torch.from_numpy(counts), n, cap).numpy()
    sent = np.zeros((n, 7, 7, 3), bool)
    for f in range(n):
        for k in range(min(cap, counts[f])):
            sent[f, recs[f, k]["row"], recs[f, k]["col"], recs[f, k]["anchor"]] = True
    assert np.array_equal(sparse.reshape(n, 7, 7, 3, 6)[sent], heads.reshape(n, 7, 7, 3, 6)[sent]) and (sparse.reshape(n, 7, 7, 3, 6)[~sent] == -128).all()
    for f in range(n):                                          # and decoding the sparse heads gives the kept records back, in order
        got = orc.decode_py(sparse[f], f, 1.0, 1.0)
        assert [tuple(g) for g in got] == [tuple(r) for r in recs[f, :min(cap, counts[f])].tolist()]


def test_c_shard_range_equals_the_python_one(yf):
    """yf_network_shard_range is what a C host application uses (multi-GPU); same split as sharding.py."""
    import ctypes
    lib = yf.load()
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    for n in (0, 1, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, b = ctypes.c_long(), ctypes.c_long()
                lib.yf_network_shard_range(n, r, world, ctypes.byref(a), ctypes.byref(b))
                assert (a.value, b.value) == sh.shard_range(n, r, world)
