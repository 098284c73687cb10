import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaprompt_amd import ops, _lib
dev = torch.device("cuda:0")
def run(B, H, N, M, d, use_mask):
    C = H * d
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B, N, C, generator=g) * 2).to(torch.bfloat16).to(dev)
    k = (torch.randn(B, M, C, generator=g) * 2).to(torch.bfloat16).to(dev)
    v = torch.randn(B, M, C, generator=g).to(torch.bfloat16).to(dev)
    km = None
    if use_mask:
        mask = (torch.rand(B, M, generator=torch.Generator().manual_seed(5)) > 0.3)
        mask[:, 0] = True
        km = mask.to(torch.uint8).to(dev).contiguous()
    _lib.call("adap_attention_set_debug", 2, -1, -1, -1)
    out, lse = ops.attention_fwd(q, k, v, H, km)
    var = _lib.call_long("adap_attention_fwd_last_variant")
    _lib.call("adap_attention_set_debug", 0, -1, -1, -1)
    out0, lse0 = ops.attention_fwd(q, k, v, H, km)
    torch.cuda.synchronize()
    e = (out.float() - out0.float()).abs()
    print(f"B{B} H{H} N{N} M{M} d{d} mask{use_mask} variant {var}: max|d out| {float(e.max()):.4f} nonfinite {int((~torch.isfinite(out.float())).sum())} "
          f"max|d lse| {float((lse - lse0).abs().max()):.5f}")
    if float(e.max()) > 0.05:
        bad = (e.amax(dim=-1) > 0.05)          # [B, N]
        rows = bad.nonzero()
        print("   bad rows:", rows.shape[0], "first", rows[:5].tolist(), "last", rows[-3:].tolist())
        eh = e.view(B, N, H, d).amax(dim=(0, 1, 3))
        print("   per head max err", [round(float(x), 3) for x in eh])
        ed = e.view(B, N, H, d).amax(dim=(0, 1, 2))
        print("   per d max err", [round(float(x), 2) for x in ed])
for case in [(2, 8, 300, 333, 40, True), (2, 8, 300, 333, 40, False), (1, 8, 256, 256, 40, False), (1, 8, 64, 64, 40, False), (1, 8, 512, 128, 40, False), (4, 8, 4096, 4096, 40, False)]:
    run(*case)
