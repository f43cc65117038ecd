import importlib, sys, numpy as np, torch, time
sys.path.insert(0, '/root/repo')
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
rng = np.random.default_rng(1)
x = rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n,7,7,18), dtype=torch.int8, device='cuda')
for nn in (4096,):
    if nn != n:
        d_in = d_in.repeat(nn//n,1,1,1).contiguous(); d_out = torch.zeros((nn,7,7,18), dtype=torch.int8, device='cuda')
    for f, w in ((1,4),(2,4),(2,8),(4,8),(2,8)):
        net.configure(f, w)
        net.time_device(d_in.data_ptr(), d_out.data_ptr(), nn, 3)
        ms = net.time_device(d_in.data_ptr(), d_out.data_ptr(), nn, 20)
        print(f"n={nn} {net.kernel_name}: {ms:.3f} ms/launch -> {nn/ms*1e3/1e6:.2f} M frames/s, {nn*10290/ms*1e3/1e9:.1f} GB/s algorithmic")
