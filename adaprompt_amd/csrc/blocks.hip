// Block-level orchestration inside the library: the launches of one UNet ResBlock (openaimodel.py:259-279), forward and data
// gradient, issued from ONE C call -- the same entry points (adap_groupnorm_fwd/bwd, adap_conv2d_nhwc), in the order and with
// the arguments of the Python mirror (functional.ResBlockFn), so the results are bit-identical to it.  Why: the host needs
// ~26 ms to issue a training step against ~31 ms on the GPU, and a box with a 10 % slower host is host-bound; a ResBlock issued
// from Python is 5-6 ctypes calls, ~13 torch.empty and ~100 us of interpreter time each way, from here it is one call.
// Frozen weights only (no weight gradients): the training path with `unfreeze_model` stays in Python.
#include "common.h"
#include "../../include/adaprompt_hip.h"

namespace {
inline int conv3(const void* x, int x_dtype, int Cin, const void* w, const float* bias, const float* chan_add, const float* residual,
                 float* y32, void* y16, int B, int H, int W, int Cout, float* sk_ws, void* stream) {
    return adap_conv2d_nhwc(x, x_dtype, Cin, w, bias, chan_add, chan_add ? Cout : 0, residual, residual ? Cout : 0, y32, y32 ? Cout : 0,
                            y16, y16 ? Cout : 0, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0, 1.0f, 0, sk_ws, 1, 0, 0, 0, 0, stream);
}
inline int conv1(const void* x, int x_dtype, int Cin, const void* w, const float* bias, float* y32, int B, int H, int W, int Cout,
                 float* sk_ws, void* stream) {
    return adap_conv2d_nhwc(x, x_dtype, Cin, w, bias, nullptr, 0, nullptr, 0, y32, Cout, nullptr, 0, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0,
                            0, 1.0f, 0, sk_ws, 1, 0, 0, 0, 0, stream);
}
}  // namespace

// x f32 [B,H,W,Cin]; emb_out f32 [B,Cout]; packs as ops.PackedConv.fwd (c1: Cin -> Cout 3x3, c2: Cout -> Cout 3x3, sk: Cin -> Cout
// 1x1 or NULL = identity, then Cin == Cout).  Writes a1 bf16 [.., Cin], h1 bf16 [.., Cout], a2 bf16 [.., Cout], skip f32 [.., Cout]
// (only with sk), out f32 [.., Cout], stats f32 [4][B][32] = mean1, rstd1, mean2, rstd2.  gn_ws / sk_ws: the larger of the two
// GroupNorms' / three contractions' workspaces (adap_groupnorm_workspace_floats / adap_conv2d_workspace_floats); gn_sync as
// adap_groupnorm_fwd.
extern "C" int adap_resblock_fwd(const float* x, const float* emb_out, const float* g1w, const float* g1b, const float* g2w,
                                 const float* g2b, const void* c1w, const float* c1b, const void* c2w, const float* c2b, const void* skw,
                                 const float* skb, void* a1, void* h1, void* a2, float* skip, float* out, float* stats, float* gn_ws,
                                 float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream) {
    ADAP_REQUIRE(x && emb_out && g1w && g1b && g2w && g2b && c1w && c2w && a1 && h1 && a2 && out && stats && gn_ws,
                 ADAP_ERR_SHAPE, "resblock_fwd: null pointer");
    ADAP_REQUIRE((skw != nullptr) == (skip != nullptr) && (skw || Cin == Cout), ADAP_ERR_SHAPE, "resblock_fwd: skip connection");
    const int HW = H * W;
    float *m1 = stats, *r1 = stats + (size_t)B * 32, *m2 = stats + (size_t)2 * B * 32, *r2 = stats + (size_t)3 * B * 32;
    int rc;
    if ((rc = adap_groupnorm_fwd(x, 0, Cin, g1w, g1b, nullptr, 0, a1, Cin, m1, r1, gn_ws, gn_sync, B, HW, Cin, 1e-5f, 1, stream))) return rc;
    if ((rc = conv3(a1, 1, Cin, c1w, c1b, emb_out, nullptr, nullptr, h1, B, H, W, Cout, sk_ws, stream))) return rc;
    if ((rc = adap_groupnorm_fwd(h1, 1, Cout, g2w, g2b, nullptr, 0, a2, Cout, m2, r2, gn_ws, gn_sync, B, HW, Cout, 1e-5f, 1, stream)))
        return rc;
    const float* res = x;
    if (skw) {
        if ((rc = conv1(x, 0, Cin, skw, skb, skip, B, H, W, Cout, sk_ws, stream))) return rc;
        res = skip;
    }
    return conv3(a2, 1, Cout, c2w, c2b, nullptr, res, out, nullptr, B, H, W, Cout, sk_ws, stream);
}

// The data gradient of the same block.  g: the gradient of `out`, f32 (g_dtype 0) or its bf16 side copy (1), plus g32 = the f32
// gradient itself (the identity skip adds it to dx).  Packs as ops.PackedConv.bwd.  Scratch: ga2 bf16 [.., Cout], gh1 bf16
// [.., Cout], ga1 bf16 [.., Cin].  Writes gx f32 [.., Cin] and gx16 = its bf16 copy.
extern "C" int adap_resblock_bwd(const void* g, int g_dtype, const float* g32, const float* x, const void* h1, const float* stats,
                                 const float* g1w, const float* g1b, const float* g2w, const float* g2b, const void* c1wb,
                                 const void* c2wb, const void* skwb, void* ga2, void* gh1, void* ga1, float* gx, void* gx16,
                                 float* gn_ws, float* sk_ws, void* gn_sync, int B, int H, int W, int Cin, int Cout, void* stream) {
    ADAP_REQUIRE(g && g32 && x && h1 && stats && g1w && g1b && g2w && g2b && c1wb && c2wb && ga2 && gh1 && ga1 && gx && gx16 && gn_ws,
                 ADAP_ERR_SHAPE, "resblock_bwd: null pointer");
    ADAP_REQUIRE(skwb || Cin == Cout, ADAP_ERR_SHAPE, "resblock_bwd: skip connection");
    const int HW = H * W;
    const float *m1 = stats, *r1 = stats + (size_t)B * 32, *m2 = stats + (size_t)2 * B * 32, *r2 = stats + (size_t)3 * B * 32;
    int rc;
    // d a2 = conv2^T g (the flipped / transposed pack: stride 1, pad K - 1 - pad = 1)
    if ((rc = conv3(g, g_dtype, Cout, c2wb, nullptr, nullptr, nullptr, nullptr, ga2, B, H, W, Cout, sk_ws, stream))) return rc;
    if ((rc = adap_groupnorm_bwd(ga2, 1, Cout, h1, 1, Cout, g2w, g2b, m2, r2, nullptr, 0, 0, gh1, Cout, nullptr, 0, gn_ws, gn_sync, B, HW,
                                 Cout, 1, stream)))
        return rc;
    if ((rc = conv3(gh1, 1, Cout, c1wb, nullptr, nullptr, nullptr, nullptr, ga1, B, H, W, Cin, sk_ws, stream))) return rc;
    if (!skwb)          // identity skip: dx + g straight into a new tensor
        return adap_groupnorm_bwd(ga1, 1, Cin, x, 0, Cin, g1w, g1b, m1, r1, gx, Cin, 1, gx16, Cin, g32, Cin, gn_ws, gn_sync, B, HW, Cin, 1,
                                  stream);
    if ((rc = conv1(g, g_dtype, Cout, skwb, nullptr, gx, B, H, W, Cin, sk_ws, stream))) return rc;
    return adap_groupnorm_bwd(ga1, 1, Cin, x, 0, Cin, g1w, g1b, m1, r1, gx, Cin, 1, gx16, Cin, nullptr, 0, gn_ws, gn_sync, B, HW, Cin, 1,
                              stream);
}
