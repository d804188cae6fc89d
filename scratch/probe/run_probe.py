import ctypes, sys, os, time
here = os.path.dirname(os.path.abspath(__file__))
mode = sys.argv[1]
if mode == "plain":
    l = ctypes.CDLL(os.path.join(here, "libprobe_sys.so"))
    l.probe_info(); print("self ->", l.probe_self())
else:
    t0=time.time(); import torch; print("import torch", time.time()-t0)
    l = ctypes.CDLL(os.path.join(here, "libprobe_sys.so"))
    l.probe_info()
    print([m.split()[-1] for m in open('/proc/self/maps') if 'amdhip' in m and 'r-xp' in m])
    a = torch.zeros(1000, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
    l.probe_run.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    print("run ->", l.probe_run(a.data_ptr(), b.data_ptr(), 1000, s)); torch.cuda.synchronize()
    print("b sum", int(b.sum()), "self ->", l.probe_self())
