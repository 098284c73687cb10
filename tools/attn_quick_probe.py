import os, sys
sys.path.insert(0, "/root/repo")
import torch
from adaprompt_amd import ops
dev = torch.device("cuda:0")
B, H, N, d = 4, 8, 4096, 40
C = H * d
qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16)
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
IT = int(os.environ.get('ADAP_PROBE_ITERS', '200'))
def timed(fn, it=None):
    it = it or IT
    for _ in range(it): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for scale in (None, 0.0):
    o, lse = ops.attention_fwd(q, k, v, H, scale=scale)
    tf = timed(lambda: ops.attention_fwd(q, k, v, H, scale=scale))
    tb = timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, scale=scale))
    print(f"env QB1={os.environ.get('ADAP_ATTN_QB1')} QSPLIT={os.environ.get('ADAP_ATTN_DKV_QSPLIT')} scale={scale}: fwd {tf:.1f} us bwd {tb:.1f} us", flush=True)
