"""Batch sharding across the GPUs of one node: frames are independent, so rank r takes the contiguous
slice [r*B/G, (r+1)*B/G) and the only exchange is ONE all-gather per step of the DETECTIONS -- fixed-capacity records plus
the true per-frame counts, packed into one buffer per rank (RCCL over xGMI when the process group backend is "nccl";
"gloo" in the CPU tests and the one-GPU rehearsal).  The reference has no multi-device path (single core);
this is the north star's design."""
import torch
import torch.distributed as dist

DET_BYTES = 28            # sizeof(yf_det), include/yf_network.h
HEAD_BYTES = 7 * 7 * 18   # one int8 head


def shard_range(n, rank, world):
    """Contiguous split; the first n % world ranks take one extra frame."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class _HostGather:
    """Pending all-gather of the rehearsal backends: the collective runs on host copies, wait() lands it in the device buffer."""

    def __init__(self, work, host_out, dev_out):
        self.work, self.host_out, self.dev_out = work, host_out, dev_out

    def wait(self):
        self.work.wait()
        self.dev_out.copy_(self.host_out)


class Slot:
    """Where one step's results go: slot `k` of buffer `i` (a buffer holds `gather_every` steps)."""
    __slots__ = ("i", "k")

    def __init__(self, i, k):
        self.i, self.k = i, k

    def __iter__(self):
        return iter((self.i, self.k))


WIRE_DET_BYTES = 12       # compact wire record: {u8 anchor, row, col, 0, i8 q[6] (the firing anchor's six quantised head values), u16 0}


def pack_compact(dets, counts, heads, n, cap):
    """Compact wire form of one step's records (device-agnostic tensor ops; `dets` uint8 [n*cap*28], `counts` int32 [n], `heads` uint8 [n*882]).
    A yf_det (28 B, include/yf_network.h) carries the frame index (= the record's position), a float confidence and four int32 box edges, all of
    which follow from the firing cell's six int8 head values through the decode tables -- so the wire carries those six bytes and the cell's
    (anchor, row, col): 12 B per record, LOSSLESS whatever the box edges are (int16 edges would not be: a wide box overflows them).  The receiver
    rebuilds a sparse head (`unpack_compact`) and runs the library's own decode on it, which gives the same records in the same order."""
    recs = dets.view(n, cap, DET_BYTES)
    anchor, row, col = (recs[..., k].long().clamp_(max=m) for k, m in ((4, 2), (5, 6), (6, 6)))     # (clamped: slots beyond the count hold stale bytes)
    valid = torch.arange(cap, device=dets.device)[None, :] < counts.clamp(max=cap)[:, None]
    f = torch.arange(n, device=dets.device)[:, None].expand(n, cap)
    q = heads.view(n, 7, 7, 3, 6)[f, row, col, anchor]                                              # [n, cap, 6] uint8 views of the int8 logits
    out = torch.zeros((n, cap, WIRE_DET_BYTES), dtype=torch.uint8, device=dets.device)
    out[..., 0], out[..., 1], out[..., 2] = anchor.to(torch.uint8), row.to(torch.uint8), col.to(torch.uint8)
    out[..., 4:10] = q
    return (out * valid[..., None].to(torch.uint8)).view(-1)


def unpack_compact(wire, counts, n, cap):
    """The sparse int8 heads [n, 7, 7, 18] a compact record block stands for: -128 everywhere (a confidence logit that never fires) except the
    six values of every transmitted (anchor, row, col).  Decoding it (yf_network_decode_device / interpreter.decode_boxes) reproduces the sender's
    records: the kept records are the first `cap` of a frame in decode order, and decoding only those keeps that order.  `counts` are the sender's
    TRUE counts (they travel beside the records and may exceed cap)."""
    w = wire.view(n, cap, WIRE_DET_BYTES)
    valid = torch.arange(cap, device=wire.device)[None, :] < counts.clamp(max=cap)[:, None]
    f = torch.arange(n, device=wire.device)[:, None].expand(n, cap)[valid]
    heads = torch.full((n, 7, 7, 3, 6), 0x80, dtype=torch.uint8, device=wire.device)
    heads[f, w[..., 1][valid].long(), w[..., 2][valid].long(), w[..., 0][valid].long()] = w[..., 4:10][valid]
    return heads.view(torch.int8).view(n, 7, 7, 18)


class DetectionExchange:
    """The per-step exchange of bench.py and of any multi-GPU caller.

    A step's record is [detection records n x cap x 28 B | counts n x 4 B (| int8 heads n x 882 B)] in ONE packed uint8 block (16-byte
    aligned sections).  A BUFFER holds `gather_every` (K) such blocks -- K consecutive steps -- and ONE all_gather_into_tensor per buffer
    sends them: K = 1 (default) is the north star's all-gather per step; K > 1 trades latency of the results for collectives (the exchange is
    latency-bound at these sizes -- 0.48 MB per rank and step --, so one collective per K steps costs one launch + handshake, not K, at K x the
    bytes).  `n_buf` buffers alternate (two at N > 1, four with two launch streams): the all-gather of a buffer is asynchronous and runs while the
    kernels of the next steps fill the others; a buffer is handed out again (`acquire`) only after `wait()` on the gather that last read it.  `compact` sends
    12-byte wire records instead of the 28-byte yf_det (`pack_compact`: 0.21 MB instead of 0.48 MB per rank and step at cap 4, lossless); the
    28-byte records stay what the kernel writes and what the C side reads.  Every rank's shard has the same size n (weak scaling: fixed-shape
    collective, no padding).

        slot = ex.acquire()                   # Slot(i = buffer, k = step within the buffer); waits for the gather that last read buffer i when k == 0
        ... launch the kernel with ex.dets_ptr(slot), ex.counts_ptr(slot), ex.heads(slot) ...
        ex.exchange(slot)                     # when the buffer's last slot is filled: (pack,) async all-gather of buffer i into ex.gathered[i]
        ex.drain()                            # a partly filled buffer is sent too; all pending gathers done

    Launch streams: acquire() / exchange() order themselves against the CURRENT torch stream (an RCCL collective waits for the work queued on
    it, `wait()` makes it wait for the collective).  A caller that launches consecutive steps on several streams passes them as
    `launch_streams`: a buffer's gather then waits for every stream that filled it, and the first launch into a re-acquired buffer on another
    stream waits for the gather too.  With K = 1 and as many buffers as streams a buffer always meets the same stream and no cross-stream wait
    is ever issued."""

    def __init__(self, n, cap, world, device, gather_heads=False, backend=None, n_buf=None, group=None, gather_every=1, compact=False, launch_streams=None, always=False, packer=None, unpacker=None):
        self.n, self.cap, self.world, self.device, self.group = n, cap, world, torch.device(device), group
        self.gather_heads, self.compact, self.K = bool(gather_heads), bool(compact), int(gather_every)
        self.active = world > 1 or bool(always)    # `always`: issue the collective in a ONE-rank group too (rehearsal of the RCCL path on a one-GPU box)
        assert self.K >= 1
        self.launch_streams = list(launch_streams) if launch_streams else []
        # compact records on a GPU: ONE launch of the library's pack kernel per step (packer(dets_ptr, counts_ptr, heads_ptr, wire_ptr, n, cap) on the current
        # stream: Network.pack_detections_device) -- the tensor-op form below is a dozen small launches, each a neighbour of the CU-filling kernel (+100 us per
        # step in the one-rank rehearsal); it stays as the statement of the format and as what the CPU (gloo) tests run
        self.packer, self.unpacker = packer, unpacker
        self.off_c = (n * cap * DET_BYTES + 15) & ~15
        self.off_h = (self.off_c + n * 4 + 15) & ~15
        self.rec_bytes = self.off_h + (((n * HEAD_BYTES + 15) & ~15) if gather_heads else 0)         # one step's block as the kernel fills it
        # one step's block on the wire: the same, or [compact records | counts (| heads)]
        self.w_off_c = (n * cap * WIRE_DET_BYTES + 15) & ~15
        self.w_off_h = (self.w_off_c + n * 4 + 15) & ~15
        self.wire_rec_bytes = (self.w_off_h + (((n * HEAD_BYTES + 15) & ~15) if gather_heads else 0)) if self.compact else self.rec_bytes
        # buffers that alternate.  A kernel waits for the gather that last read ITS buffer; on a GPU that the kernels fill, that gather only gets to run when the
        # step behind its own drains -- with two launch streams and two buffers every kernel then waits for a gather that is waiting for the kernel in front
        # of it (+13 us per step in the one-rank rehearsal, profiles/r05_exchange_one_rank.txt); with twice as many buffers as streams the gather a kernel waits
        # for is four steps old and long done (+0.7 us).  (A multiple of the stream count: a buffer then always meets the same stream.)
        self.n_buf = n_buf if n_buf is not None else (max(2, 2 * len(self.launch_streams)) if self.active else max(1, len(self.launch_streams)))
        self.backend = backend or (dist.get_backend(group) if self.active else "none")
        dev, K = self.device, self.K
        self.local = [torch.zeros((K * self.rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(self.n_buf)]
        self._heads = [[self._block(self.local[i], k)[self.off_h:self.off_h + n * HEAD_BYTES] if gather_heads
                        else torch.zeros((n * HEAD_BYTES,), dtype=torch.uint8, device=dev) for k in range(K)] for i in range(self.n_buf)]
        self.wire = [torch.zeros((K * self.wire_rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(self.n_buf)] if (self.compact and self.active) else self.local
        self.gathered = [torch.zeros((world * K * self.wire_rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(self.n_buf)] if self.active else []
        self.pending = [None] * self.n_buf
        self.filled = [0] * self.n_buf      # slots of the buffer filled since it was acquired
        self.step_no = 0
        self.waits = 0                      # gathers a later acquire() had to wait for (bookkeeping, tests)
        self.collectives = 0                # all-gathers issued (bookkeeping, tests)

    # ---- buffers
    def _block(self, buf, k, rec_bytes=None):
        rb = rec_bytes or self.rec_bytes
        return buf[k * rb:(k + 1) * rb]

    @staticmethod
    def _ik(slot):
        return (slot.i, slot.k) if isinstance(slot, Slot) else (int(slot), 0)

    def dets_ptr(self, slot):
        i, k = self._ik(slot)
        return self.local[i].data_ptr() + k * self.rec_bytes

    def counts_ptr(self, slot):
        return self.dets_ptr(slot) + self.off_c

    def heads(self, slot):
        """int8 heads of a slot as a uint8 tensor [n * 882] (inside the record when gather_heads, a side buffer otherwise)"""
        i, k = self._ik(slot)
        return self._heads[i][k]

    def views(self, buf, r=0, k=0):
        """(records [n, cap, 28] uint8, counts [n] int32) of rank r's step-k block inside `buf` (a local buffer: r = 0; or a gathered one of
        the 28-byte form)"""
        base = self._block(buf, r * self.K + k)
        return base[:self.n * self.cap * DET_BYTES].view(self.n, self.cap, DET_BYTES), base[self.off_c:self.off_c + self.n * 4].view(torch.int32)

    def wire_views(self, buf, r=0, k=0):
        """(wire records, counts [n] int32) of rank r's step-k block of a WIRE buffer (self.wire[i] with r = 0, or self.gathered[i]): the records
        are uint8 [n, cap, 12] in the compact form, [n, cap, 28] otherwise"""
        if not self.compact:
            return self.views(buf, r, k)
        base = self._block(buf, r * self.K + k, self.wire_rec_bytes)
        return base[:self.n * self.cap * WIRE_DET_BYTES].view(self.n, self.cap, WIRE_DET_BYTES), base[self.w_off_c:self.w_off_c + self.n * 4].view(torch.int32)

    # ---- the double-buffer protocol
    def _other_streams(self):
        if not self.launch_streams or self.device.type != "cuda":
            return []
        cur = torch.cuda.current_stream(self.device)
        return [s for s in self.launch_streams if s != cur]

    def acquire(self):
        i, k = (self.step_no // self.K) % self.n_buf, self.step_no % self.K
        self.step_no += 1
        if k == 0:
            if self.pending[i] is not None:   # the gather that last read this buffer must be done before it is overwritten
                self.pending[i].wait()
                self.pending[i] = None
                self.waits += 1
                if self.K > 1:                # later slots of this buffer are filled from other streams: they wait for the gather as well
                    cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
                    for s in self._other_streams():
                        s.wait_stream(cur)
            self.filled[i] = 0
        return Slot(i, k)

    def _send(self, i):
        """(pack and) all-gather buffer i: its first filled[i] slots are fresh, the collective always has the fixed K-slot shape"""
        assert self.pending[i] is None, "buffer exchanged twice without acquire()"
        if self.K > 1 or len(self.launch_streams) > self.n_buf:
            cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
            for s in self._other_streams():  # the collective is ordered behind the current stream: bring in the streams that filled the other slots
                cur.wait_stream(s)
        if self.compact:
            for k in range(self.filled[i]):
                blk, wblk = self._block(self.local[i], k), self._block(self.wire[i], k, self.wire_rec_bytes)
                counts = blk[self.off_c:self.off_c + self.n * 4].view(torch.int32)
                if self.packer is not None:
                    self.packer(blk.data_ptr(), blk.data_ptr() + self.off_c, self._heads[i][k].data_ptr(), wblk.data_ptr(), self.n, self.cap)
                else:
                    wblk[:self.n * self.cap * WIRE_DET_BYTES] = pack_compact(blk[:self.n * self.cap * DET_BYTES], counts, self._heads[i][k], self.n, self.cap)
                wblk[self.w_off_c:self.w_off_c + self.n * 4] = blk[self.off_c:self.off_c + self.n * 4]
                if self.gather_heads:
                    wblk[self.w_off_h:self.w_off_h + self.n * HEAD_BYTES] = self._heads[i][k]
        self.collectives += 1
        if self.backend == "nccl":            # RCCL on the device buffers, on RCCL's own stream
            self.pending[i] = dist.all_gather_into_tensor(self.gathered[i], self.wire[i], group=self.group, async_op=True)
        else:                                 # rehearsal: the same collective over host copies of the same buffers
            host_in = self.wire[i].cpu()
            host_out = torch.empty((self.world * self.K * self.wire_rec_bytes,), dtype=torch.uint8)
            work = dist.all_gather_into_tensor(host_out, host_in, group=self.group, async_op=True)
            self.pending[i] = _HostGather(work, host_out, self.gathered[i])

    def exchange(self, slot):
        i, k = self._ik(slot)
        self.filled[i] = k + 1
        if self.active and k == self.K - 1:
            self._send(i)

    def drain(self):
        if self.active:
            for i in range(self.n_buf):       # a buffer the run ended in the middle of: its filled slots are sent (fixed-shape collective)
                if self.pending[i] is None and 0 < self.filled[i] < self.K:
                    self._send(i)
        for i in range(self.n_buf):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
            self.filled[i] = 0 if self.active else self.filled[i]
        # a run that ended inside a buffer: the next acquire() starts a FRESH buffer (slot 0).  Round 5 left step_no in the middle of the sent buffer, so a
        # caller that kept stepping was handed slot k != 0 of it, and the buffer's next send packed range(filled) = the stale slots 0..k-1 in front of the
        # fresh one -- retransmitted, and indistinguishable from fresh blocks on the receiver.
        self.step_no = -(-self.step_no // self.K) * self.K

    # ---- what every rank must hold after exchange(slot) + drain(): every rank's blocks, in rank (= frame) order
    def check_gathered(self, slot, rank):
        """True if gathered[i] holds this rank's wire block of the slot at its place and the counts of all ranks form one [world * n] array."""
        i, k = self._ik(slot)
        mine = self._block(self.gathered[i], rank * self.K + k, self.wire_rec_bytes)
        ok = bool(torch.equal(mine, self._block(self.wire[i], k, self.wire_rec_bytes)))
        return ok and tuple(self.gathered_counts(slot).shape) == (self.world * self.n,)

    def gathered_counts(self, slot):
        i, k = self._ik(slot)
        return torch.cat([self.wire_views(self.gathered[i], r, k)[1] for r in range(self.world)])

    def gathered_records(self, slot):
        """every rank's wire records of the slot, rank-major: [world * n, cap, 28] (or [.., 12] in the compact form)"""
        i, k = self._ik(slot)
        return torch.cat([self.wire_views(self.gathered[i], r, k)[0] for r in range(self.world)])

    def rank_major_samples(self, slot, per_rank=64):
        """Global frame indices to spot-check the rank = frame order of a gathered slot with: the first `per_rank` FIRING frames of EVERY rank's shard
        (round 5 looked at the first 512 firing frames of the whole gathered array -- all of them inside rank 0's shard at N = 8, so a block of rank >= 1
        landing at the wrong place would have passed)."""
        counts = self.gathered_counts(slot).cpu().numpy()
        out = []
        for r in range(self.world):
            firing = (counts[r * self.n:(r + 1) * self.n] > 0).nonzero()[0][:per_rank]
            out += [int(r * self.n + f) for f in firing]
        return out

    def gathered_is_rank_major(self, slot, rank, local_counts, frame_of_record, per_rank=64):
        """What every rank must hold after the exchange of `slot`: its own block at ITS place, the counts of all ranks as one [world * n] array with this rank's
        counts in [rank * n, (rank + 1) * n), and -- for firing frames sampled from every rank's shard -- `frame_of_record(g)` (the frame index carried by the
        first gathered record of global frame g) == g mod n: a rank's records carry LOCAL frame indices, the global order is rank-major.
        Returns (ok, ranks whose shard contributed a sample)."""
        ok = self.check_gathered(slot, rank)
        g_counts = self.gathered_counts(slot)
        a, b = rank * self.n, (rank + 1) * self.n
        ok = ok and tuple(g_counts.shape) == (self.world * self.n,) and bool(torch.equal(g_counts[a:b].cpu(), local_counts.cpu()))
        samples = self.rank_major_samples(slot, per_rank)
        ok = ok and all(int(frame_of_record(g)) == g % self.n for g in samples)
        return bool(ok), sorted({g // self.n for g in samples})

    def gathered_sparse_heads(self, slot):
        """compact form: the sparse int8 heads [world * n, 7, 7, 18] the gathered records of the slot stand for (decode them to get yf_det records)"""
        i, k = self._ik(slot)
        assert self.compact
        if self.unpacker is not None:          # unpacker(wire_ptr, counts_ptr, heads_ptr, n, cap): Network.unpack_detections_device, one launch per rank's block
            out = torch.empty((self.world * self.n, 7, 7, 18), dtype=torch.int8, device=self.device)
            for r in range(self.world):
                w, c = self.wire_views(self.gathered[i], r, k)
                self.unpacker(w.data_ptr(), c.data_ptr(), out[r * self.n:(r + 1) * self.n].data_ptr(), self.n, self.cap)
            return out
        return torch.cat([unpack_compact(*self.wire_views(self.gathered[i], r, k), self.n, self.cap) for r in range(self.world)])
