import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_heads = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
cap = 16
d_dets = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda"); d_counts = torch.zeros((n,), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream(); sp = stream.cuda_stream
def run(K, events, decode):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        if events: evs[k][0].record(stream)
        net.run_device(d_in.data_ptr(), d_heads.data_ptr(), n, sp)
        if events: evs[k][1].record(stream)
        if decode: net.decode_device(d_heads.data_ptr(), n, d_dets.data_ptr(), d_counts.data_ptr(), cap, yf.YF_DECODE_PY, 1.0, 1.0, sp)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    km = np.mean([a.elapsed_time(b) for a, b in evs]) * 1e3 if events else float("nan")
    return el / K * 1e6, km
for _ in range(2):
    for events in (True, False):
        for decode in (True, False):
            run(5, events, decode)
            us, km = run(40, events, decode)
            print(f"events={events!s:5} decode={decode!s:5}  {us:7.1f} us/step   kernel(event) {km:7.1f} us")
