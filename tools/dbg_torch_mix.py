import sys, os, ctypes
sys.path.insert(0,'grail-rs_amd')
order = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
def load_mine():
    import grail_hip as G
    if mode == "deepbind":
        G._lib = None
        L = ctypes.CDLL(G.LIB_PATH, mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    G.load()
    return G
if order == "mine_first":
    G = load_mine(); print("mine loaded; devices:", G.device_count())
    import torch; print("torch imported; cuda avail:", torch.cuda.is_available())
    print("mine again:", G.device_count())
    try:
        c = G.Context(0); print("ctx ok"); c.close()
    except Exception as e: print("ctx fail", e)
elif order == "torch_first":
    import torch; print("torch imported")
    G = load_mine(); print("mine devices:", G.device_count())
    try:
        c = G.Context(0); print("ctx ok"); c.close()
    except Exception as e: print("ctx fail", e)
    print("torch cuda avail:", torch.cuda.is_available())
elif order == "torch_first_noinit_query":
    import torch
    G = load_mine()
    try:
        c = G.Context(0); print("ctx ok"); 
        x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
        c.close()
    except Exception as e: print("fail", e)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "amdhip64" in l or "hsa-runtime" in l or "rccl" in l})
print(libs)
