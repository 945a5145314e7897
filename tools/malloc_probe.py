import torch, time, ctypes
torch.cuda.init(); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
def t_malloc(gb):
    p = ctypes.c_void_p()
    t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * 2**30))); hip.hipDeviceSynchronize(); t1 = time.time()
    rc2 = hip.hipMemset(p, 0, ctypes.c_size_t(int(gb * 2**30))); hip.hipDeviceSynchronize(); t2 = time.time()
    hip.hipFree(p); hip.hipDeviceSynchronize(); t3 = time.time()
    print(f"hipMalloc {gb:6.1f} GB: {t1-t0:.3f} s (rc {rc}); memset {t2-t1:.3f} s; hipFree {t3-t2:.3f} s", flush=True)
for gb in (1, 8, 32, 32, 64, 128, 32):
    t_malloc(gb)
