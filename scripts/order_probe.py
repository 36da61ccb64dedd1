"""Which of libccgp.so and PyTorch-ROCm has to be loaded first in a process that uses both:
python scripts/order_probe.py lib_first | import_torch_then_lib_then_cuda | torch_first"""
import sys
sys.path.insert(0, '.')
order = sys.argv[1]
import ccgp_amd
from ccgp_amd import api
if order == "lib_first":
    h = api.Handle(0)
    import torch
    try:
        torch.zeros(4, device="cuda")
        print("lib_first: torch ok")
    except Exception as e:
        print("lib_first: torch failed:", str(e)[:80])
elif order == "import_torch_then_lib_then_cuda":
    import torch
    h = api.Handle(0)
    try:
        torch.zeros(4, device="cuda")
        print("import torch, Handle, then cuda init: ok")
    except Exception as e:
        print("import torch, Handle, then cuda init: failed:", str(e)[:80])
else:
    import torch
    torch.zeros(4, device="cuda")
    h = api.Handle(0)
    print("torch_first: ok")
