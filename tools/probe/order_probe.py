"""Load order of libyf_network.so and PyTorch in one process (tests/test_gpu_parity.py::test_library_and_pytorch_in_either_order).  DEV TOOL."""
import sys, importlib, os, faulthandler, functools
# One run of the lib_init_first order stalled somewhere behind 'init ok' in round 3 (gpurun_out/r03_h: killed after 300 s, block-buffered output).  Every
# print is flushed and the process dumps all threads' stacks and exits by itself after 240 s, so that one occurrence names the frame it hangs in.
faulthandler.dump_traceback_later(240, exit=True)
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
yf = importlib.import_module("stm32h7-yolo_amd")
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l})
if mode == 'lib_first':
    yf.load(); print('after lib:', maps())
    import torch; print('after torch:', maps()); print('cuda available', torch.cuda.is_available())
    try: yf.Network(device=0).init(); print('init ok')
    except Exception as e: print('init failed:', str(e)[-80:])
elif mode == 'lib_init_first':
    net = yf.Network(device=0).init(); print('init ok', maps())
    import torch; print('cuda available', torch.cuda.is_available(), maps())
    x = torch.zeros(4, device='cuda'); print('torch tensor ok')
else:
    import torch; print('cuda available', torch.cuda.is_available(), maps())
    yf.Network(device=0).init(); print('init ok', maps())
