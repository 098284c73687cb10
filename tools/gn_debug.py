import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaprompt_amd import ops, _lib
dev = torch.device("cuda:0")
B, C, H = 4, 320, 64
x = (torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(1)) * 1.5 + 0.3).to(dev)
g = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2))).to(dev)
b = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(3))).to(dev)
gy = torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(4)).to(dev)
ad = torch.randn(B, H, H, C, generator=torch.Generator().manual_seed(5)).to(dev)
for mode in ("single", "two"):
    if mode == "two":
        ops.gn_two_pass(True)
    for xdt, gdt in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)):
        xi, gi = x.to(xdt), gy.to(gdt)
        y32, y16, m, r = ops.groupnorm_fwd(xi, g, b, 1e-5, 1, out_f32=True, out_bf16=True)
        v = _lib.call_long("adap_groupnorm_last_variant")
        dx32, dx16 = ops.groupnorm_bwd(gi, xi, g, b, m, r, 1, out_bf16=True, add_from=ad)
        _, dx16o = ops.groupnorm_bwd(gi, xi, g, b, m, r, 1, out_f32=False, out_bf16=True)
        torch.cuda.synchronize()
        for name, t in (("y32", y32), ("y16", y16), ("dx32", dx32), ("dx16", dx16), ("dx16o", dx16o)):
            bad = ~torch.isfinite(t.float())
            n = int(bad.sum())
            msg = f"{mode} v{v} {xdt} {gdt} {name}: nonfinite {n}"
            if n:
                idx = bad.nonzero()
                msg += f" first {idx[0].tolist()} last {idx[-1].tolist()} rows {sorted(set((idx[:,1]*H+idx[:,2]).tolist()))[:8]} ch {sorted(set(idx[:,3].tolist()))[:12]}"
            print(msg, flush=True)
        print("   y16 vs y32", float((y16.float() - y32).abs().max()), "dx16 vs dx32", float((dx16.float() - dx32).abs().max()))

print("---- exactness of the bf16 copies (single-launch path), 5 repetitions")
ops.gn_two_pass(False)
for rep in range(5):
    y32, y16, m, r = ops.groupnorm_fwd(x, g, b, 1e-5, 1, out_f32=True, out_bf16=True)
    dx32, dx16 = ops.groupnorm_bwd(gy.to(torch.bfloat16), x, g, b, m, r, 1, out_bf16=True, add_from=ad)
    torch.cuda.synchronize()
    for name, a32, a16 in (("y", y32, y16), ("dx", dx32, dx16)):
        want = a32.to(torch.bfloat16)
        bad = (want.view(torch.int16) != a16.view(torch.int16))
        n = int(bad.sum())
        msg = f"rep {rep} {name}16 != bf16({name}32): {n}"
        if n:
            idx = bad.nonzero()[:6]
            for i in idx.tolist():
                bb, yy, xx, cc = i
                msg += f"\n    at b{bb} row {yy * H + xx} ch {cc} (e={cc % 8}): f32 {float(a32[bb, yy, xx, cc]):+.6f} bits32 {a32[bb, yy, xx, cc].view(torch.int32).item() & 0xffffffff:08x} got16 {a16[bb, yy, xx, cc].view(torch.int16).item() & 0xffff:04x} want16 {want[bb, yy, xx, cc].view(torch.int16).item() & 0xffff:04x}"
        print(msg, flush=True)
