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


class DetectionExchange:
    """The per-step exchange of bench.py and of any multi-GPU caller.

    A rank's record is [detection records n x cap x 28 B | counts n x 4 B (| int8 heads n x 882 B)] in ONE packed uint8 buffer
    (16-byte aligned sections), so a step issues ONE all_gather_into_tensor.  `n_buf` such buffers alternate (two at N > 1):
    the all-gather of step k is asynchronous and runs while the kernel of step k+1 fills the other buffer; a buffer is handed
    out again (`acquire`) only after `wait()` on the gather that last read it.  Every rank's shard has the same size n (weak
    scaling: fixed-shape collective, no padding).

        i = ex.acquire()                      # buffer index for this step; waits for the gather that last read it
        ... launch the kernel with ex.dets_ptr(i), ex.counts_ptr(i), ex.heads(i) ...
        ex.exchange(i)                        # async all-gather of buffer i into ex.gathered[i]
        ex.drain()                            # all pending gathers done
    """

    def __init__(self, n, cap, world, device, gather_heads=False, backend=None, n_buf=None, group=None):
        self.n, self.cap, self.world, self.device, self.group = n, cap, world, torch.device(device), group
        self.gather_heads = bool(gather_heads)
        self.off_c = (n * cap * DET_BYTES + 15) & ~15
        self.off_h = (self.off_c + n * 4 + 15) & ~15
        self.rec_bytes = self.off_h + (((n * HEAD_BYTES + 15) & ~15) if gather_heads else 0)
        self.n_buf = n_buf if n_buf is not None else (2 if world > 1 else 1)
        self.backend = backend or (dist.get_backend(group) if world > 1 else "none")
        dev = self.device
        self.local = [torch.zeros((self.rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(self.n_buf)]
        self._heads = [self.local[i][self.off_h:self.off_h + n * HEAD_BYTES] if gather_heads
                       else torch.zeros((n * HEAD_BYTES,), dtype=torch.uint8, device=dev) for i in range(self.n_buf)]
        self.gathered = [torch.zeros((world * self.rec_bytes,), dtype=torch.uint8, device=dev) for _ in range(self.n_buf)] if world > 1 else []
        self.pending = [None] * self.n_buf
        self.step_no = 0
        self.waits = 0                      # gathers a later acquire() had to wait for (bookkeeping, tests)

    # ---- buffers
    def dets_ptr(self, i):
        return self.local[i].data_ptr()

    def counts_ptr(self, i):
        return self.local[i].data_ptr() + self.off_c

    def heads(self, i):
        """int8 heads of buffer i as a uint8 tensor [n * 882] (inside the record when gather_heads, a side buffer otherwise)"""
        return self._heads[i]

    def views(self, buf, r=0):
        """(records [n, cap, 28] uint8, counts [n] int32) of rank r's record inside `buf` (a local or a gathered buffer)"""
        base = buf[r * self.rec_bytes:(r + 1) * self.rec_bytes]
        return base[:self.n * self.cap * DET_BYTES].view(self.n, self.cap, DET_BYTES), base[self.off_c:self.off_c + self.n * 4].view(torch.int32)

    # ---- the double-buffer protocol
    def acquire(self):
        i = self.step_no % self.n_buf
        self.step_no += 1
        if self.pending[i] is not None:       # the gather that last read this buffer must be done before it is overwritten
            self.pending[i].wait()
            self.pending[i] = None
            self.waits += 1
        return i

    def exchange(self, i):
        if self.world == 1:
            return
        assert self.pending[i] is None, "buffer exchanged twice without acquire()"
        if self.backend == "nccl":            # RCCL on the device buffers, on RCCL's own stream
            self.pending[i] = dist.all_gather_into_tensor(self.gathered[i], self.local[i], group=self.group, async_op=True)
        else:                                 # rehearsal: the same collective over host copies of the same buffers
            host_in = self.local[i].cpu()
            host_out = torch.empty((self.world * self.rec_bytes,), dtype=torch.uint8)
            work = dist.all_gather_into_tensor(host_out, host_in, group=self.group, async_op=True)
            self.pending[i] = _HostGather(work, host_out, self.gathered[i])

    def drain(self):
        for i in range(self.n_buf):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None

    # ---- what every rank must hold after exchange(i) + drain(): every rank's record, in rank (= frame) order
    def check_gathered(self, i, rank):
        """True if gathered[i] holds this rank's record at its slot and the counts of all ranks form one [world * n] array."""
        g = self.gathered[i]
        ok = bool(torch.equal(g[rank * self.rec_bytes:(rank + 1) * self.rec_bytes], self.local[i]))
        counts = torch.cat([self.views(g, r)[1] for r in range(self.world)])
        return ok and tuple(counts.shape) == (self.world * self.n,)

    def gathered_counts(self, i):
        return torch.cat([self.views(self.gathered[i], r)[1] for r in range(self.world)])

    def gathered_records(self, i):
        return torch.cat([self.views(self.gathered[i], r)[0] for r in range(self.world)])
