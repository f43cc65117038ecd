"""Host wait policy against a 20-step timed region (the driver's flags): hipSetDeviceFlags 1 = spin, 2 = yield, 4 = blocking sync, -1 = leave the default.
Measured in round 4: default / spin / yield 139.3 us per step (= the kernel time), blocking +1.3 %: nothing to gain.  DEV TOOL.  usage: sync_wait_probe.py <mode>"""
import ctypes, importlib, os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mode = int(sys.argv[1])
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
if mode >= 0:
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(mode)); print("hipSetDeviceFlags", mode, "->", rc)
yf = importlib.import_module("stm32h7-yolo_amd")
n, cap = 4096, 4
net = yf.Network().init()
rng = np.random.default_rng(5)
ins = [torch.from_numpy(rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)).cuda() for _ in range(8)]
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_dets = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"); d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
def run(k): net.run_decode_device(ins[k % 8].data_ptr(), d_out.data_ptr(), n, d_dets.data_ptr(), d_cnt.data_ptr(), cap)
for k in range(600): run(k)
torch.cuda.synchronize()
ts = []
for rep in range(60):
    for k in range(5): run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20): run(k)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20 * 1e6)
print(f"mode {mode}: wall clock per step over 20-step regions: median {np.median(ts):.2f} us, min {min(ts):.2f}")
