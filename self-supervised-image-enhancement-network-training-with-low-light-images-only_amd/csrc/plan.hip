// Plan executor: the whole hot path of the reference as a static schedule of HIP launches.
//
//   ssie_plan_enhance_fwd   == LowLightEnhance.forward            (/root/reference/model.py:229-234)
//   ssie_plan_loss_fwd_bwd  == compute_loss + loss.backward()     (model.py:544-575, :315)
//   ssie_adam_step          == torch.optim.Adam.step              (model.py:213, :316)
//
// A plan is built once per (N, bands, H, W); binding it to a caller-owned workspace and to the flat
// parameter / gradient buffers materialises every launch descriptor, so a step is a fixed sequence of
// kernel launches with no host-side shape logic, no allocation and no synchronisation.
#include "layer_ops.h"
#include "loss_kernels.h"
#include "attention.h"
#include "spectral_conv.h"
#include "../../include/ssie_hip.h"
#include <functional>
#include <map>
#include <string>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <math.h>

namespace {

typedef std::function<int(hipStream_t)> FnT;
// op kinds for per-kernel-class profiling (bench.py roofline): see ssie_plan_profile_step
enum { K_FPROP2 = 0, K_FPROP1, K_WGRAD, K_WGRAD_REDUCE, K_COLSUM, K_PACK, K_LOSS, K_FFT, K_ATTN, K_ELEMENTWISE, K_SPEC, K_WINO, K_WGRAD_WINO, K_WINO4, K_NKINDS };
// How a launch touches the two weight-gradient slab areas (run_ops_overlapped orders the streams by THIS, never by kind):
//   SLAB_WRITE  produces partial slabs in area `slab` on the main stream (any weight-gradient kernel)
//   SLAB_READ   consumes area `slab` on the side stream (slab reduction, Winograd tap extraction - anything reading the area)
enum { SLAB_NONE = 0, SLAB_WRITE = 1, SLAB_READ = 2 };
struct Fn {
    FnT fn; int kind; double flops; std::string tag;
    double bytes = 0.0;  // algorithmic HBM bytes of the launch: every operand tensor read once, every result written once (0 = not stated)
    int slab = 0;       // which of the two slab areas
    int slab_use = SLAB_NONE;
    Fn(FnT f, int k = K_ELEMENTWISE, double fl = 0.0, std::string t = "", int sl = 0, int use = SLAB_NONE)
        : fn(std::move(f)), kind(k), flops(fl), tag(std::move(t)), slab(sl), slab_use(use) {}
    int operator()(hipStream_t st) const { return fn(st); }
};

struct ParamInfo { std::string name; size_t off; int shape[4]; int ndim; size_t numel; };
struct BufInfo { size_t off; int N, H, W, C, cs; };
struct LayerP { size_t w, b; int cout, cin, k; bool transposed; };

const int CH = 64;
const int kWgs = 512;
int g_spectral9 = 1;     // ssie_debug_set_spectral9: 0 = the 9 x 9 convolution on the direct MFMA kernels (plans created afterwards)

struct Plan {
    int N, B, H, W, CX, CRL;
    int H2, W2, H4, W4, H8, W8;
    std::vector<ParamInfo> params;
    size_t nparam_floats = 0;
    std::map<std::string, BufInfo> bufs;
    size_t ws_floats = 0;
    size_t partial_off = 0, packdesc_off = 0, mask_off = 0, scal_off = 0;
    size_t lpart_off = 0, fpart_off = 0, counter_off = 0, fftws_off = 0, tailw_off = 0, skinny_off = 0;
    // frequency-domain 9 x 9 convolution (spectral_conv.hip): tiles per pass, padded band count, buffers (float offsets)
    bool spectral = false; int sp_Mt = 0, sp_Kp = 0, sp_slices = 2;      // 544 frequencies x 2 slices = 1 088 reduction workgroups
    size_t sp_Xf = 0, sp_Yf = 0, sp_Zf = 0, sp_Gn = 0, sp_Bf = 0, sp_Bd = 0, sp_dW = 0;
    int counter_cursor = 0;
    int loss_blocks = 0, fft_blocks = 0; size_t fftws_floats = 0;
    float coefs[8];
    // bound state
    float* ws = nullptr; float* P = nullptr; float* G = nullptr;
    std::vector<PackDesc> packs;
    size_t pack_cursor = 0, pack_floats_total = 0, pack_off = 0, pack_cap = 0;    // pack_cap: what the dry build reserved (packed weights + slab regions)
    std::vector<Fn> fwd, pass2, lossbwd;
    std::vector<Fn> fwd16;       // enhance-only forward with bf16 storage / bf16 MFMA (ssie_plan_enhance_fwd_bf16)
    std::vector<Fn> fwdi;        // enhance-only forward in fp32: `fwd` with its last three launches replaced by the fused tail
    size_t npacks_train = 0;     // packs[0 .. npacks_train) belong to the fp32 lists, the rest to fwd16
    bool bound = false;
    bool fold_masks = true;      // captured at creation (ssie_debug_set_fold_masks): ReLU / sigmoid mask launches folded into the producing launches
    bool qkv_fused = true;       // captured at creation (ssie_debug_set_qkv_fused): the dry build and the bound build must walk the same packs
    bool fused_tail = true;      // captured at creation (ssie_debug_set_fused_tail): the dry build and the bound build must agree
    // the slab reductions (HBM-bound) run on a side stream underneath the next MFMA-bound launches; wgrad launches
    // alternate between two slab areas so that a reduction only has to finish before the wgrad AFTER the next one
    int slab_seq = 0;
    hipStream_t side = nullptr;
    // hipGraph of one train step (ssie_debug_set_graph): captured on cap_stream at the second call, replayed on the caller's stream
    hipStream_t cap_stream = nullptr; hipGraphExec_t gexec = nullptr; int train_calls = 0; bool graph_failed = false, use_graph = false;
    hipEvent_t ev_w[2] = {nullptr, nullptr}, ev_r[2] = {nullptr, nullptr};
    ~Plan() {
        for (int i = 0; i < 2; ++i) { if (ev_w[i]) hipEventDestroy(ev_w[i]); if (ev_r[i]) hipEventDestroy(ev_r[i]); }
        if (side) hipStreamDestroy(side);
        if (gexec) hipGraphExecDestroy(gexec);
        if (cap_stream) hipStreamDestroy(cap_stream);
    }

    float* buf(const char* n) { return ws + bufs.at(n).off; }
    const BufInfo& bi(const char* n) { return bufs.at(n); }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void add_param(Plan& pl, const std::string& name, int d0, int d1 = 0, int d2 = 0, int d3 = 0)
{
    ParamInfo pi; pi.name = name; pi.off = pl.nparam_floats;
    pi.shape[0] = d0; pi.shape[1] = d1; pi.shape[2] = d2; pi.shape[3] = d3;
    pi.ndim = d3 ? 4 : (d1 ? 2 : 1);
    pi.numel = (size_t)d0 * (d1 ? d1 : 1) * (d2 ? d2 : 1) * (d3 ? d3 : 1);
    pl.nparam_floats = align_up(pl.nparam_floats + pi.numel, 4);
    pl.params.push_back(pi);
}

// state-dict order of the reference module (model.py:33-47, 93-97, 125-141)
void build_params(Plan& pl)
{
    const int B = pl.B, c = CH;
    auto cw = [&](const std::string& n, int co, int ci, int k) { add_param(pl, n + ".weight", co, ci, k, k); add_param(pl, n + ".bias", co); };
    const std::string d = "decomposition_net.", i = "illum_adjust_net.";
    cw(d + "conv0.0", c / 2, B, 3); cw(d + "shallow_conv.0", c, B, 9); cw(d + "conv1.0", c, c, 3);
    cw(d + "conv2.0", 2 * c, c, 3); cw(d + "conv3.0", 2 * c, 2 * c, 3);
    add_param(pl, d + "deconv.0.weight", 2 * c, c, 3, 3); add_param(pl, d + "deconv.0.bias", c);
    cw(d + "conv5.0", c, 2 * c, 3); cw(d + "conv7.0", c, c + c / 2, 3); cw(d + "recon", B + 1, c, 3);
    cw(i + "conv0.0", c, B + 1, 3); cw(i + "conv1.0", c, c, 3); cw(i + "conv2.0", c, c, 3); cw(i + "conv3.0", c, c, 3);
    const char* lin[5] = {"q_linear", "k_linear", "v_linear", "ff_linear1", "ff_linear2"};
    for (int q = 0; q < 5; ++q) { add_param(pl, i + "attn." + lin[q] + ".weight", 64, 64); add_param(pl, i + "attn." + lin[q] + ".bias", 64); }
    cw(i + "deconv1.0", c, c, 3); cw(i + "deconv2.0", c, c, 3); cw(i + "deconv3.0", c, c, 3);
    cw(i + "feature_fusion.0", c, 3 * c, 1); cw(i + "final_conv", 1, c, 3);
}

LayerP layer(Plan& pl, const std::string& name, bool transposed = false)
{
    LayerP L; memset(&L, 0, sizeof(L));
    for (size_t q = 0; q < pl.params.size(); ++q) {
        if (pl.params[q].name == name + ".weight") {
            const ParamInfo& w = pl.params[q];
            L.w = w.off; L.b = pl.params[q + 1].off; L.transposed = transposed;
            L.k = w.ndim == 4 ? w.shape[2] : 1;
            if (transposed) { L.cin = w.shape[0]; L.cout = w.shape[1]; } else { L.cout = w.shape[0]; L.cin = w.shape[1]; }
            return L;
        }
    }
    return L;
}

size_t alloc(Plan& pl, const char* name, int N, int H, int W, int C)
{
    BufInfo b; b.off = pl.ws_floats; b.N = N; b.H = H; b.W = W; b.C = C; b.cs = ssie_round_up(C, 4);
    pl.ws_floats = align_up(pl.ws_floats + (size_t)N * H * W * b.cs, 64);
    pl.bufs[name] = b;
    return b.off;
}

void build_buffers(Plan& pl)
{
    const int N = pl.N, H = pl.H, W = pl.W, B = pl.B;
    const int H2 = pl.H2, W2 = pl.W2, H4 = pl.H4, W4 = pl.W4, H8 = pl.H8, W8 = pl.W8;
    // Every tensor of the two decomposition passes is allocated as an adjacent [pass 1 ; pass 2] pair, i.e. ONE tensor of
    // 2N patches: the weight gradients of both passes are then a single launch per layer over the doubled batch (half the
    // wgrad launches and half the slab traffic of the shared decomposition weights).  x | S is such a pair too.
    auto alloc2 = [&](const char* n1, const char* n2, int h, int w, int c) {
        const size_t o1 = alloc(pl, n1, N, h, w, c);
        pl.ws_floats = o1 + (size_t)N * h * w * ssie_round_up(c, 4);             // no alignment gap inside the pair
        alloc(pl, n2, N, h, w, c);
    };
    alloc2("x", "S", H, W, B); alloc(pl, "gS", N, H, W, B);
    alloc(pl, "D", N, H, W, 1); alloc(pl, "gD", N, H, W, 1);
    alloc(pl, "gRL", N, H, W, B + 1);
    alloc2("c0_1", "c0_2", H, W, 32); alloc2("sh_1", "sh_2", H, W, 64); alloc2("c1_1", "c1_2", H, W, 64);
    alloc2("c2_1", "c2_2", H2, W2, 128); alloc2("c3_1", "c3_2", H2, W2, 128); alloc2("dc_1", "dc_2", H, W, 64);
    alloc2("c5_1", "c5_2", H, W, 64); alloc2("c7_1", "c7_2", H, W, 64); alloc2("RL_1", "RL_2", H, W, B + 1);
    // decomposition gradients (w.r.t. pre-activation): pass 1 under the plain name, pass 2 as <name>_2 right behind it
    alloc2("G8", "G8_2", H, W, B + 1); alloc2("G7", "G7_2", H, W, 64); alloc2("G5", "G5_2", H, W, 64); alloc2("G0", "G0_2", H, W, 32);
    alloc2("Gdc", "Gdc_2", H, W, 64); alloc2("G3", "G3_2", H2, W2, 128); alloc2("G2", "G2_2", H2, W2, 128);
    alloc2("G1", "G1_2", H, W, 64); alloc2("Gsh", "Gsh_2", H, W, 64);
    // illumination net
    alloc(pl, "a0", N, H, W, 64); alloc(pl, "a1", N, H2, W2, 64); alloc(pl, "a2", N, H4, W4, 64); alloc(pl, "a3", N, H8, W8, 64);
    alloc(pl, "qkv", N, H8, W8, 192); alloc(pl, "ao", N, H8, W8, 64); alloc(pl, "f1", N, H8, W8, 64); alloc(pl, "t3", N, H8, W8, 64);
    alloc(pl, "lse", N, 4, H8 * W8, 1); alloc(pl, "delta", N, 4, H8 * W8, 1);
    alloc(pl, "u1", N, H4, W4, 64); alloc(pl, "d1", N, H4, W4, 64); alloc(pl, "u2", N, H2, W2, 64); alloc(pl, "d2", N, H2, W2, 64);
    alloc(pl, "u3", N, H, W, 64); alloc(pl, "d3", N, H, W, 64); alloc(pl, "f", N, H, W, 64);
    alloc(pl, "Gf", N, H, W, 64); alloc(pl, "Gf2", N, H2, W2, 64); alloc(pl, "Gf4", N, H4, W4, 64); alloc(pl, "gd3", N, H, W, 64); alloc(pl, "Ge3", N, H, W, 64); alloc(pl, "tmpH", N, H, W, 64);
    alloc(pl, "gd2", N, H2, W2, 64); alloc(pl, "Ge2", N, H2, W2, 64); alloc(pl, "gd1", N, H4, W4, 64); alloc(pl, "Ge1", N, H4, W4, 64);
    alloc(pl, "gt3", N, H8, W8, 64); alloc(pl, "gf1", N, H8, W8, 64); alloc(pl, "gao", N, H8, W8, 64); alloc(pl, "gqkv", N, H8, W8, 192);
    // bf16 inference path: bf16 copies of the tensors that also exist in fp32 (allocated in float units: C/2)
    alloc(pl, "xh", N, H, W, ssie_round_up(B, 8) / 2); alloc(pl, "RLh", N, H, W, ssie_round_up(B + 1, 8) / 2); alloc(pl, "aoh", N, H8, W8, 32);
    // scratch
    // (the weight-gradient slab regions follow the packed weights: one per launch, sized by the dry build - Builder::take_slabs)
    pl.partial_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + 256 * 256, 64);
    pl.loss_blocks = 2048; pl.fft_blocks = ssie_fft_partials(N, B, H, W);
    pl.lpart_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + (size_t)pl.loss_blocks * 8, 64);
    pl.fpart_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + (size_t)pl.fft_blocks, 64);
    pl.scal_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + 16, 64);
    pl.fftws_floats = ssie_fft_workspace_floats(N, B, H, W);
    pl.fftws_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + pl.fftws_floats, 64);   // three-pass Fourier loss (band-grouped rows, or planes larger than the LDS)
    pl.mask_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + ((size_t)H * W + 3) / 4, 64);
    pl.tailw_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + ssie_tail_weight_floats(), 64);     // composite weights of the fused tail
    {   // spectral 9 x 9 (tile-major): X^ of both passes [2 Mt][f][Kp], Y^ / halo-G^ [Mt][f][64], Z^ [Mt][f][Kp], no-halo G^ [2 Mt][f][64], weights, dW^ slices
        int ty, tx; pl.sp_Mt = N * ssie_spec_tiles(H, W, &ty, &tx); pl.sp_Kp = pl.CX;
        pl.spectral = g_spectral9 && pl.CX % 32 == 0;
        if (pl.spectral) {
            const size_t nf = SSIE_SPEC_NF, Mt = pl.sp_Mt, Kp = pl.sp_Kp;
            auto take = [&](size_t complexes) { const size_t o = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + 2 * complexes, 64); return o; };
            pl.sp_Xf = take(nf * 2 * Mt * Kp); pl.sp_Yf = take(nf * Mt * 64); pl.sp_Zf = take(nf * Mt * Kp); pl.sp_Gn = take(nf * 2 * Mt * 64);
            pl.sp_Bf = take(nf * Kp * 64); pl.sp_Bd = take(nf * 64 * Kp); pl.sp_dW = take(((size_t)pl.sp_slices * nf + 9 * SSIE_SPEC_KX) * Kp * 64);
        }
    }
    pl.skinny_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + ssie_skinny_wgrad_ws_floats(), 64);   // final_conv weight-gradient partials
    pl.counter_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + 1024, 64);      // tile-queue counters, one per conv launch
    pl.packdesc_off = pl.ws_floats; pl.ws_floats = align_up(pl.ws_floats + 384 * sizeof(PackDesc) / 4, 64);
    pl.pack_off = pl.ws_floats;     // packed weights grow from here at bind time (size known after a dry build)
}

// -------------------------------------------------------------------------------------------------
extern int g_overlap, g_batched_reduce;

struct Builder {
    Plan& pl;
    bool dry;                 // dry run: only count packed-weight floats
    bool h16 = false;         // building the bf16 inference list: bf16 sources / packs / kernels
    explicit Builder(Plan& p, bool d) : pl(p), dry(d) { pl.pack_cursor = 0; pl.packs.clear(); pl.counter_cursor = 0; }

    float* take_pack(size_t floats)
    {
        float* r = dry ? nullptr : pl.ws + pl.pack_off + pl.pack_cursor;
        pl.pack_cursor = align_up(pl.pack_cursor + floats, 64);
        return r;
    }
    // packed-weight floats of a conv launch: room for either form when the Winograd kernel may take it
    static size_t pack_floats(int K, int N, int k, int stride)
    {
        const size_t a = ssie_packed_floats(K, N, k * k);
        const size_t w = (k == 3 && stride == 1) ? ssie_wino_packed_floats(K, N) : 0;
        return a > w ? a : w;
    }
    SrcDesc src(const char* name, int C, int Hv, int Wv, int coff = 0)
    {
        if (h16) {
            // bf16 tensors live in the fp32 tensors' allocations (first half); the three that must also exist in fp32
            // (input cube, R/I output, attention output) have bf16 twins.  Strides are in ELEMENTS either way.
            const char* hn = !strcmp(name, "x") ? "xh" : !strcmp(name, "RL_1") ? "RLh" : !strcmp(name, "ao") ? "aoh" : name;
            const BufInfo& b = pl.bi(hn);
            const int cs = hn != name ? b.cs * 2 : b.cs;
            return ssie_make_src(dry ? nullptr : pl.buf(hn), ssie_round_up(C, 8), ssie_round_up(cs, 8), coff, b.H, b.W, Hv, Wv);
        }
        const BufInfo& b = pl.bi(name);
        return ssie_make_src(dry ? nullptr : pl.buf(name), C, b.cs, coff, b.H, b.W, Hv, Wv);
    }
    float* ptr(const char* name) { return dry ? nullptr : pl.buf(name); }
    float* par(size_t off) { return dry ? nullptr : pl.P + off; }
    float* grad(size_t off) { return dry ? nullptr : pl.G + off; }
    void push(std::vector<Fn>& ops, ConvParams p, int k_real)
    {
        if (dry) return;
        if (pl.counter_cursor < 1024) p.tile_counter = (int*)(pl.ws + pl.counter_off) + pl.counter_cursor++;
        // algorithmic FLOPs: real (un-padded) channels and taps only
        const double fl = 2.0 * p.N * p.Ho * p.Wo * (double)p.Cout * k_real * p.ntaps;
        char tag[96];
        snprintf(tag, sizeof(tag), "%sconv k%d->n%d taps%d si%d so%d %dx%d", p.wino == 2 ? "winograd F(4x4,3x3) " : p.wino ? "winograd " : p.tconv ? "transposed (4 classes) " : "", k_real, p.Cout, p.ntaps, p.si, p.so, p.Ho, p.Wo);
        // algorithmic bytes: each source's physical pixels (an up-sampled source is read at ITS resolution) x the channels taken, the
        // output positions of this launch x real output channels, the fused epilogue operands, the packed weights
        const double ein = h16 ? 2.0 : 4.0, eout = (h16 && p.out_bf16) ? 2.0 : 4.0;
        double by = 0.0;
        for (int s = 0; s < p.nsrc; ++s) {
            const double hs = p.src[s].Hs < p.Hv ? p.src[s].Hs : p.Hv, ws = p.src[s].Ws < p.Wv ? p.src[s].Ws : p.Wv;
            by += (double)p.N * hs * ws * p.src[s].C * ein;
        }
        const double opos = (double)p.N * (p.tconv ? 4.0 : 1.0) * p.Ho * p.Wo;
        by += opos * p.Cout * eout * (p.accumulate ? 2.0 : 1.0);
        if (p.out2) by += opos * p.Cout * (h16 ? 2.0 : 4.0);
        if (p.addsrc) by += opos * p.Cout * ein;
        if (p.mask_y) by += opos * p.Cout * 4.0;
        by += (double)k_real * p.Cout * p.ntaps * ein;
        if (h16) {
            std::string t16 = std::string("bf16 ") + tag;
            ops.push_back(Fn([p](hipStream_t st) { return ssie_launch_fprop_bf16(p, st); }, p.Cout_pad % 64 == 0 ? K_FPROP2 : K_FPROP1, fl, t16));
            ops.back().bytes = by;
            return;
        }
        ops.push_back(Fn([p](hipStream_t st) { return ssie_launch_fprop(p, st); }, p.wino == 2 ? K_WINO4 : p.wino ? K_WINO : p.Cout_pad % 64 == 0 ? K_FPROP2 : K_FPROP1, fl, tag));
        ops.back().bytes = by;
    }
    // bf16 list: which outputs stay fp32 (API outputs and the attention operands)
    static bool out_is_f32(const char* out) { return !strcmp(out, "RL_1") || !strcmp(out, "qkv") || !strcmp(out, "D"); }

    // forward conv (stride 1/2) over concatenated / up-sampled sources
    int conv(std::vector<Fn>& ops, const LayerP& L, std::vector<SrcDesc> srcs, int Hv, int Wv, int stride,
             const char* out, int act, const char* addsrc = nullptr, const char* out2 = nullptr, int out_coff = 0)
    {
        const int T = L.k * L.k, pad = (L.k - 1) / 2;
        TapList t = ssie_taps_conv(L.k);
        float* wp = take_pack(pack_floats(L.cin, L.cout, L.k, stride));
        if (dry) return 0;
        const BufInfo& ob = pl.bi(out);
        Epilogue e; memset(&e, 0, sizeof(e)); e.bias = pl.P + L.b; e.act = act;
        e.addsrc = addsrc ? pl.buf(addsrc) : nullptr; e.out2 = out2 ? pl.buf(out2) : nullptr;
        const int Ho = (Hv + 2 * pad - L.k) / stride + 1, Wo = (Wv + 2 * pad - L.k) / stride + 1;
        if (Ho != ob.H || Wo != ob.W) return SSIE_E_SHAPE;
        ConvParams p;
        if (h16) {
            pl.packs.push_back(ssie_make_pack_bf16(pl.P + L.w, wp, L.cin, L.cout, t, T, L.cin * T, 1));
            // the pre-skip copies (u1..u3) only serve the backward pass; the R/I output gets its bf16 twin instead
            e.out2 = !strcmp(out, "RL_1") ? pl.buf("RLh") : nullptr;
            const int f32o = out_is_f32(out);
            int rc = ssie_make_conv_bf16(p, srcs.data(), (int)srcs.size(), pl.N, Hv, Wv, t, stride, Ho, Wo, wp, L.cout,
                                         pl.buf(out), !f32o, ob.H, ob.W, f32o ? ob.cs : ssie_round_up(ob.cs, 8), out_coff, 1, 0, 0, e);
            if (rc) return rc;
            if (e.out2) p.out2_cstride = pl.bi("RLh").cs * 2;      // the twin's own pixel stride (B + 1 padded to 8 bf16 elements)
            push(ops, p, L.cin);
            return 0;
        }
        int rc = ssie_make_conv(p, srcs.data(), (int)srcs.size(), pl.N, Hv, Wv, t, stride, Ho, Wo, wp, L.cout,
                                pl.buf(out), ob.H, ob.W, ob.cs, out_coff, 1, 0, 0, e);
        if (rc) return rc;
        if (const int wk = ssie_wino_eligible(p, t)) {
            pl.packs.push_back(ssie_make_pack_wino(pl.P + L.w, wp, L.cin, L.cout, t, T, L.cin * T, 1, wk));
            ssie_conv_to_wino(p, wp, wk);
        } else pl.packs.push_back(ssie_make_pack(pl.P + L.w, wp, L.cin, L.cout, t, T, L.cin * T, 1));
        push(ops, p, L.cin);
        return 0;
    }

    // transposed-conv machinery: ConvTranspose2d forward (weight (in,out,3,3)) and dgrad of a stride-2 conv (OIHW)
    int transposed(std::vector<Fn>& ops, const float* wbase, int Kc, int Nc, int s_k, int s_n, SrcDesc in, int Hin, int Win,
                   const char* out, const Epilogue& e)
    {
        const BufInfo& ob = pl.bi(out);
        // all four output-parity classes in one launch where the input fills the chip (conv_tconv.hip); the pack is reserved
        // either way so that the dry run and the real build walk the same cursor
        float* wpm = take_pack(ssie_packed_floats(Kc, Nc, 9));
        const bool merged = !dry && !h16 && ssie_tconv_eligible(in, pl.N, Hin, Win, Nc) && ob.H <= 2 * Hin && ob.W <= 2 * Win;
        if (merged) {
            TapList t = ssie_taps_transposed_all();
            pl.packs.push_back(ssie_make_pack(wbase, wpm, Kc, Nc, t, s_k, s_n, 1));
            ConvParams p;
            int rc = ssie_make_conv(p, &in, 1, pl.N, Hin, Win, t, 1, Hin, Win, wpm, Nc, pl.buf(out), ob.H, ob.W, ob.cs, 0, 2, 0, 0, e);
            if (rc) return rc;
            ssie_conv_to_tconv(p);
            push(ops, p, Kc);
        }
        for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
            TapList t = ssie_taps_transposed(3, 1, py, px);
            float* wp = take_pack(ssie_packed_floats(Kc, Nc, t.n));
            if (dry || merged) continue;
            ConvParams p;
            const int Ho = ssie_ceil_div(ob.H - py, 2), Wo = ssie_ceil_div(ob.W - px, 2);
            if (h16) {
                pl.packs.push_back(ssie_make_pack_bf16(wbase, wp, Kc, Nc, t, s_k, s_n, 1));
                int rc = ssie_make_conv_bf16(p, &in, 1, pl.N, Hin, Win, t, 1, Ho, Wo, wp, Nc, pl.buf(out), 1, ob.H, ob.W, ob.cs, 0, 2, py, px, e);
                if (rc) return rc;
                push(ops, p, Kc);
                continue;
            }
            pl.packs.push_back(ssie_make_pack(wbase, wp, Kc, Nc, t, s_k, s_n, 1));
            int rc = ssie_make_conv(p, &in, 1, pl.N, Hin, Win, t, 1, Ho, Wo, wp, Nc, pl.buf(out), ob.H, ob.W, ob.cs, 0, 2, py, px, e);
            if (rc) return rc;
            push(ops, p, Kc);
        }
        return 0;
    }

    // out2 / out2_mode (ConvParams.out2_mode): 1 = out receives the masked gradient and out2 the unmasked one; 2 = out accumulates the
    // unmasked total and out2 receives the masked total - a ReLU / sigmoid mask launch folded into the producing data gradient
    Epilogue bwd_epi(const char* mask_y, int mask_mode, int accumulate, const char* out2 = nullptr, int out2_mode = 0)
    {
        Epilogue e; memset(&e, 0, sizeof(e));
        e.mask_y = (mask_y && !dry) ? pl.buf(mask_y) : nullptr; e.mask_mode = mask_y ? mask_mode : 0; e.accumulate = accumulate;
        e.out2 = (out2 && !dry) ? pl.buf(out2) : nullptr; e.out2_mode = out2 ? out2_mode : 0;
        return e;
    }

    // data gradient of a forward conv layer w.r.t. input channels [ci_off, ci_off + cs)
    int dgrad(std::vector<Fn>& ops, const LayerP& L, int stride, const char* g, int g_coff, int ci_off, int cs,
              const char* gx, const char* mask_y, int mask_mode, int accumulate, const char* out2 = nullptr, int out2_mode = 0)
    {
        const int T = L.k * L.k;
        const BufInfo& gb = pl.bi(g); const BufInfo& xb = pl.bi(gx);
        SrcDesc in = ssie_make_src(ptr(g), ssie_round_up(L.cout, 4), gb.cs, g_coff, gb.H, gb.W, gb.H, gb.W);
        Epilogue e = bwd_epi(mask_y, mask_mode, accumulate, out2, out2_mode);
        if (out2 && (stride != 1 || pl.bi(out2).cs != xb.cs || pl.bi(out2).H != xb.H || pl.bi(out2).W != xb.W)) return SSIE_E_ARG;
        const float* wbase = dry ? nullptr : pl.P + L.w + (size_t)ci_off * T;
        if (stride == 1) {
            TapList t = ssie_taps_dgrad_s1(L.k);
            float* wp = take_pack(pack_floats(L.cout, cs, L.k, 1));
            if (dry) return 0;
            ConvParams p;
            int rc = ssie_make_conv(p, &in, 1, pl.N, gb.H, gb.W, t, 1, xb.H, xb.W, wp, cs, pl.buf(gx), xb.H, xb.W, xb.cs, 0, 1, 0, 0, e);
            if (rc) return rc;
            if (const int wk = ssie_wino_eligible(p, t)) {
                pl.packs.push_back(ssie_make_pack_wino(wbase, wp, L.cout, cs, t, L.cin * T, T, 1, wk));
                ssie_conv_to_wino(p, wp, wk);
            } else pl.packs.push_back(ssie_make_pack(wbase, wp, L.cout, cs, t, L.cin * T, T, 1));
            push(ops, p, L.cout);
            return 0;
        }
        return transposed(ops, wbase, L.cout, cs, L.cin * T, T, in, gb.H, gb.W, gx, e);
    }

    // weight gradient (+ fused bias gradient when with_bias) of a forward conv layer, one input source per call
    // nbatch = 2: x and g name the pass-1 halves of [pass 1 ; pass 2] pairs and the launch covers both passes
    int wgrad(std::vector<Fn>& ops, const LayerP& L, int stride, SrcDesc x, int creal, int Hv, int Wv, int ci_off, const char* g,
              int g_coff = 0, bool with_bias = false, int nbatch = 1, int co_group = 0, long w_extra = 0, long b_extra = 0)
    {
        const int T = L.k * L.k, pad = (L.k - 1) / 2;
        const BufInfo& gb = pl.bi(g);
        const int Ho = (Hv + 2 * pad - L.k) / stride + 1, Wo = (Wv + 2 * pad - L.k) / stride + 1;
        if (Ho != gb.H || Wo != gb.W) return SSIE_E_SHAPE;
        TapList t = ssie_taps_conv(L.k);
        WgradParams p;
        int rc = ssie_make_wgrad(p, x, pl.N * nbatch, Hv, Wv, 0, ptr(g), gb.cs, g_coff, L.cout, Ho, Wo, stride, t, nullptr, kWgs);
        if (rc) return rc;
        const size_t need = ssie_wgrad_slab_floats(p);
        const size_t bneed = with_bias ? (size_t)p.nslices * p.co_pad : 0;
        float* slabs = take_slabs(need + bneed);                     // this launch's own region (the dry run sizes it)
        if (dry) return 0;
        p.slabs = slabs;
        const int sl = pl.slab_seq++ & 1;
        float* dw = pl.G + L.w + (size_t)ci_off * T;
        const long s_co = (long)L.cin * T;
        const int cout = L.cout;
        float* bslab = with_bias ? slabs + need : nullptr;
        float* db = with_bias ? pl.G + L.b : nullptr;
        p.bias_slabs = bslab;
        const double fl = 2.0 * pl.N * nbatch * Ho * Wo * (double)cout * creal * T;
        char tag[96];
        snprintf(tag, sizeof(tag), "%swgrad ci%d co%d taps%d si%d %dx%d slices%d", p.wino ? "winograd " : "", creal, cout, T, stride, Ho, Wo, p.nslices);
        ops.push_back(Fn([p](hipStream_t st) { return ssie_launch_wgrad(p, st); }, p.wino ? K_WGRAD_WINO : K_WGRAD, fl, tag, sl, SLAB_WRITE));
        {   // input at its physical resolution + output gradient, read once each; partial slabs written once ...
            const double hs = x.Hs < Hv ? x.Hs : Hv, ws = x.Ws < Wv ? x.Ws : Wv;
            ops.back().bytes = 4.0 * ((double)pl.N * nbatch * (hs * ws * creal + (double)Ho * Wo * cout) + (double)need + (double)bneed);
        }
        // ... and read once by the reduction (dW read-modify-write)
        reduce(ops, ssie_make_reduce(slabs, p.nslices, p.ntaps, p.ci_pad, p.co_pad, creal, cout, dw, s_co, T, 1, bslab, db, 1, -1, co_group, w_extra, b_extra),
               4.0 * ((double)need + (double)bneed + 2.0 * (double)creal * cout * T), sl);
        return 0;
    }

    // Each weight-gradient launch owns its slab region (sized by the dry run, behind the packed weights: ~0.3 GB at the 31-band
    // configuration - nothing on 288 GB), so no reduction has to finish before a later launch may write: the reductions of one backward
    // pass are collected and run as ONE batched launch at its end (flush_reduces; g_batched_reduce = 0 or the side-stream executor:
    // one launch per layer right behind its producer, as before)
    float* take_slabs(size_t floats) { return take_pack(floats); }
    std::vector<ReduceDesc> pending;
    double pending_bytes = 0.0;
    void reduce(std::vector<Fn>& ops, const ReduceDesc& d, double bytes, int sl)
    {
        if (g_batched_reduce && !g_overlap) { pending.push_back(d); pending_bytes += bytes; return; }
        ops.push_back(Fn([d](hipStream_t st) { return ssie_launch_wgrad_reduce(d.slabs, d.nslices, d.ntaps, d.ci_pad, d.co_pad, d.Cin, d.Cout, d.dst, d.s_co, d.s_ci, d.s_t,
                                                                             d.bias_slabs, d.db, d.accumulate, st, d.accumulate_bias, d.co_group, d.w_extra, d.b_extra); },
                         K_WGRAD_REDUCE, 0.0, "", sl, SLAB_READ));
        ops.back().bytes = bytes;
    }
    void flush_reduces(std::vector<Fn>& ops)
    {
        for (size_t at = 0; at < pending.size(); at += SSIE_REDUCE_BATCH) {
            const size_t n = pending.size() - at < SSIE_REDUCE_BATCH ? pending.size() - at : SSIE_REDUCE_BATCH;
            std::vector<ReduceDesc> part(pending.begin() + at, pending.begin() + at + n);
            char tag[64]; snprintf(tag, sizeof(tag), "batched reduction of %zu layers", n);
            // slab = 2: reads every region; it runs in launch order behind all producers (no side stream in this mode)
            ops.push_back(Fn([part](hipStream_t st) { return ssie_launch_wgrad_reduce_batched(part.data(), (int)part.size(), st); }, K_WGRAD_REDUCE, 0.0, tag, 2, SLAB_READ));
            ops.back().bytes = pending_bytes * (double)n / (double)pending.size();
        }
        pending.clear(); pending_bytes = 0.0;
    }

    // q_linear | k_linear | v_linear (model.py:93-95, 104-106) as one 64 -> 192 1 x 1 layer into "qkv": three sub-block weight packs and
    // three bias copies fill one packed operand / one contiguous bias vector (the three parameter tensors are not adjacent)
    int qkv_fwd(std::vector<Fn>& ops)
    {
        const std::string i = "illum_adjust_net.attn.";
        const LayerP L3[3] = {layer(pl, i + "q_linear"), layer(pl, i + "k_linear"), layer(pl, i + "v_linear")};
        TapList t = ssie_taps_conv(1);
        float* wp = take_pack(ssie_packed_floats(64, 192, 1));
        float* bp = take_pack(192);
        if (dry) return 0;
        for (int j = 0; j < 3; ++j) {
            PackDesc d = ssie_make_pack(pl.P + L3[j].w, wp, 64, 64, t, 1, 64, 1);
            d.Npad = 192; d.n_off = 64 * j; d.ncnt = 64;
            pl.packs.push_back(d);
            PackDesc c; memset(&c, 0, sizeof(c)); c.copy = 1; c.w = pl.P + L3[j].b; c.dst = bp; c.N = 64; c.n_off = 64 * j;
            pl.packs.push_back(c);
        }
        const BufInfo& ob = pl.bi("qkv");
        Epilogue e; memset(&e, 0, sizeof(e)); e.bias = bp; e.act = ACT_NONE;
        SrcDesc in = src("a3", 64, pl.H8, pl.W8);
        ConvParams p;
        int rc = ssie_make_conv(p, &in, 1, pl.N, pl.H8, pl.W8, t, 1, pl.H8, pl.W8, wp, 192, pl.buf("qkv"), ob.H, ob.W, ob.cs, 0, 1, 0, 0, e);
        if (rc) return rc;
        push(ops, p, 64);
        return 0;
    }
    // data gradient of the same layer: gt3 += mask(a3) * W_qkv^T gqkv, K = 192 gradient channels from the three weight tensors
    int qkv_dgrad(std::vector<Fn>& ops)
    {
        const std::string i = "illum_adjust_net.attn.";
        const LayerP L3[3] = {layer(pl, i + "q_linear"), layer(pl, i + "k_linear"), layer(pl, i + "v_linear")};
        TapList t = ssie_taps_dgrad_s1(1);
        float* wp = take_pack(ssie_packed_floats(192, 64, 1));
        if (dry) return 0;
        for (int j = 0; j < 3; ++j) {
            PackDesc d = ssie_make_pack(pl.P + L3[j].w, wp, 64, 64, t, 64, 1, 1);      // K = output channel of the layer, N = its input channel
            d.k_off = 64 * j; d.ncnt = 64;
            pl.packs.push_back(d);
        }
        const BufInfo& gb = pl.bi("gqkv"); const BufInfo& xb = pl.bi("gt3");
        SrcDesc in = ssie_make_src(pl.buf("gqkv"), 192, gb.cs, 0, gb.H, gb.W, gb.H, gb.W);
        Epilogue e = bwd_epi("a3", MASK_RELU, 1);
        ConvParams p;
        int rc = ssie_make_conv(p, &in, 1, pl.N, gb.H, gb.W, t, 1, xb.H, xb.W, wp, 64, pl.buf("gt3"), xb.H, xb.W, xb.cs, 0, 1, 0, 0, e);
        if (rc) return rc;
        push(ops, p, 192);
        return 0;
    }

    void bias_grad(std::vector<Fn>& ops, const LayerP& L, const char* g, int g_coff = 0, int nbatch = 1)
    {
        if (dry) return;
        const BufInfo& gb = pl.bi(g);
        const float* gp = pl.buf(g); float* part = pl.ws + pl.partial_off; float* db = pl.G + L.b;
        const long npix = (long)gb.N * nbatch * gb.H * gb.W; const int cs = gb.cs, C = L.cout;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_colsum(gp, npix, cs, g_coff, C, part, C <= 128 ? 512 : 256, db, 1, st); }, K_COLSUM));
    }

    void mask_axpy(std::vector<Fn>& ops, const char* src, const char* y, int mode, const char* dst, int C, int accumulate)
    {
        if (dry) return;
        const BufInfo& sb = pl.bi(src); const BufInfo& db = pl.bi(dst);
        const float* sp = pl.buf(src); const float* yp = y ? pl.buf(y) : nullptr; float* dp = pl.buf(dst);
        const int ycs = y ? pl.bi(y).cs : 0, scs = sb.cs, dcs = db.cs;
        const long npix = (long)sb.N * sb.H * sb.W;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_mask_axpy(sp, scs, yp, ycs, mode, dp, dcs, npix, C, accumulate, st); }, K_ELEMENTWISE, 0.0, std::string("mask_axpy ") + src + "->" + dst));
        ops.back().bytes = 4.0 * npix * C * (2.0 + (y ? 1.0 : 0.0) + (accumulate ? 1.0 : 0.0));
    }

    // ---- frequency-domain 9 x 9 convolution (shallow_conv), spectral_conv.hip ----
    float2* spc(size_t off) { return dry ? nullptr : (float2*)(pl.ws + off); }
    // forward: xin (x or S) -> out (sh_1 / sh_2); pass = 1 also transforms the weights
    void spec_fwd(std::vector<Fn>& ops, const LayerP& L, const char* xin, int pass, const char* out)
    {
        if (dry) return;
        const int N = pl.N, H = pl.H, W = pl.W, Mt = pl.sp_Mt, Kp = pl.sp_Kp, cx = pl.CX, B = pl.B, m0 = pass == 1 ? 0 : Mt;
        float2* Xf = spc(pl.sp_Xf); float2* Yf = spc(pl.sp_Yf); float2* Bf = spc(pl.sp_Bf); float2* Bd = spc(pl.sp_Bd);
        const float* w = pl.P + L.w; const float* bias = pl.P + L.b; const float* xp = pl.buf(xin); float* op = pl.buf(out);
        const int ocs = pl.bi(out).cs;
        if (pass == 1) ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_weights(w, 64, B, Kp, 64, Bf, Bd, st); }, K_SPEC, 0.0, "spectral 9x9: weights"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_fft(xp, cx, Kp, N, H, W, 1, Xf, m0, 2 * Mt, st); }, K_SPEC, 0.0, "spectral 9x9: fft in"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_gemm(Xf, 2 * Mt, m0, Bf, Yf, Mt, 0, Mt, Kp, 64, st); }, K_SPEC,
                         2.0 * N * H * W * 64.0 * B * 81, "spectral 9x9: fwd gemm"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_ifft(Yf, 0, Mt, 64, N, H, W, op, ocs, 64, bias, 0, st); }, K_SPEC, 0.0, "spectral 9x9: ifft out"));
    }
    // data gradient of pass 2: g (64 ch) -> gx += ...
    void spec_dgrad(std::vector<Fn>& ops, const char* g, const char* gx)
    {
        if (dry) return;
        const int N = pl.N, H = pl.H, W = pl.W, Mt = pl.sp_Mt, Kp = pl.sp_Kp, B = pl.B;
        float2* Yf = spc(pl.sp_Yf); float2* Zf = spc(pl.sp_Zf); float2* Bd = spc(pl.sp_Bd);
        const float* gp = pl.buf(g); float* xp = pl.buf(gx); const int gcs = pl.bi(g).cs, xcs = pl.bi(gx).cs;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_fft(gp, gcs, 64, N, H, W, 1, Yf, 0, Mt, st); }, K_SPEC, 0.0, "spectral 9x9: fft grad (halo)"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_gemm(Yf, Mt, 0, Bd, Zf, Mt, 0, Mt, 64, Kp, st); }, K_SPEC,
                         2.0 * N * H * W * 64.0 * B * 81, "spectral 9x9: dgrad gemm"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_ifft(Zf, 0, Mt, Kp, N, H, W, xp, xcs, B, nullptr, 1, st); }, K_SPEC, 0.0, "spectral 9x9: ifft dgrad"));
    }
    // weight gradient over both passes: g = pass-1 half of the [Gsh ; Gsh_2] pair, X^ of x and S still in the workspace
    int spec_wgrad(std::vector<Fn>& ops, const LayerP& L, const char* g)
    {
        if (dry) return 0;
        // the bias-gradient launch stages ceil(2 Mt / 64) * 64 partial sums in the shared 256 x 256-float scratch: many small patches
        // (4 tiles per 25 x 25 patch) can exceed it while still inside the 32-bit tensor guard
        if (((size_t)2 * pl.sp_Mt + 63) / 64 * 64 > (size_t)256 * 256) return SSIE_E_WORKSPACE;
        const int N = pl.N, H = pl.H, W = pl.W, Mt = pl.sp_Mt, Kp = pl.sp_Kp, B = pl.B, ns = pl.sp_slices;
        float2* Xf = spc(pl.sp_Xf); float2* Gn = spc(pl.sp_Gn); float2* dWs = spc(pl.sp_dW);
        const float* gp = pl.buf(g); const int gcs = pl.bi(g).cs; float* dw = pl.G + L.w;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_fft(gp, gcs, 64, 2 * N, H, W, 0, Gn, 0, 2 * Mt, st); }, K_SPEC, 0.0, "spectral 9x9: fft grad (tiles)"));
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_wgrad(Xf, Gn, dWs, 2 * Mt, Kp, ns, 64, B, dw, st); }, K_SPEC,
                         2.0 * 2 * N * H * W * 64.0 * B * 81, "spectral 9x9: wgrad"));
        // the layer's bias gradient from the DC bins of the same spectra (instead of a column-sum pass over the gradient tensor)
        float* part = pl.ws + pl.partial_off; float* db = pl.G + L.b;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_spec_bias(Gn, 2 * Mt, part, db, 1, st); }, K_COLSUM, 0.0, "spectral 9x9: bias gradient (DC bins)"));
        return 0;
    }

    // mask_y / masked: also write masked = relu'(mask_y) * (the total left in dst) - the mask launch that used to follow
    void upadj(std::vector<Fn>& ops, const char* src, int Hv, int Wv, const char* dst, int accumulate, const char* mask_y = nullptr,
               const char* masked = nullptr)
    {
        if (dry) return;
        const BufInfo& db = pl.bi(dst);
        const float* sp = pl.buf(src); float* dp = pl.buf(dst);
        const int scs = pl.bi(src).cs, dcs = db.cs, Hs = db.H, Ws = db.W, N = pl.N;
        const float* yp = mask_y ? pl.buf(mask_y) : nullptr; float* mp = masked ? pl.buf(masked) : nullptr;
        const int ycs = mask_y ? pl.bi(mask_y).cs : 0, mcs = masked ? pl.bi(masked).cs : 0;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_upsample_adjoint(sp, Hv, Wv, scs, dp, Hs, Ws, dcs, N, 64, accumulate, st, yp, ycs, mp, mcs); },
                         K_ELEMENTWISE, 0.0, masked ? "upsample_adjoint + relu mask" : "upsample_adjoint"));
        ops.back().bytes = 4.0 * N * 64.0 * ((double)Hv * Wv + (double)Hs * Ws * ((accumulate ? 2.0 : 1.0) + (masked ? 2.0 : 0.0)));
    }
};

#define CK(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int g_fold_masks = 1;    // ssie_debug_set_fold_masks: 0 = the backward's activation masks as separate mask_axpy launches (plans created afterwards)
int g_qkv_fused = 1;     // ssie_debug_set_qkv_fused: 0 = q / k / v as three 64 -> 64 launches each way (plans created afterwards)
int g_skinny_final = 1;  // ssie_debug_set_skinny_final: 0 = final_conv (64 -> 1) on the MFMA tile kernels like every other layer

int build_decomposition_fwd(Builder& b, std::vector<Fn>& ops, const char* xin, int p)
{
    Plan& pl = b.pl;
    const int H = pl.H, W = pl.W, H2 = pl.H2, W2 = pl.W2;
    auto nm = [&](const char* s) { return std::string(s) + (p == 1 ? "_1" : "_2"); };
    const std::string c0 = nm("c0"), sh = nm("sh"), c1 = nm("c1"), c2 = nm("c2"), c3 = nm("c3"), dc = nm("dc"), c5 = nm("c5"), c7 = nm("c7"), RL = nm("RL");
    const std::string d = "decomposition_net.";
    CK(b.conv(ops, layer(pl, d + "conv0.0"), {b.src(xin, pl.CX, H, W)}, H, W, 1, c0.c_str(), ACT_RELU));
    if (pl.spectral && !b.h16) b.spec_fwd(ops, layer(pl, d + "shallow_conv.0"), xin, p, sh.c_str());
    else CK(b.conv(ops, layer(pl, d + "shallow_conv.0"), {b.src(xin, pl.CX, H, W)}, H, W, 1, sh.c_str(), ACT_NONE));
    CK(b.conv(ops, layer(pl, d + "conv1.0"), {b.src(sh.c_str(), 64, H, W)}, H, W, 1, c1.c_str(), ACT_RELU));
    CK(b.conv(ops, layer(pl, d + "conv2.0"), {b.src(c1.c_str(), 64, H, W)}, H, W, 2, c2.c_str(), ACT_RELU));
    CK(b.conv(ops, layer(pl, d + "conv3.0"), {b.src(c2.c_str(), 128, H2, W2)}, H2, W2, 1, c3.c_str(), ACT_RELU));
    {
        LayerP L = layer(pl, d + "deconv.0", true);
        Epilogue e; memset(&e, 0, sizeof(e)); e.bias = b.par(L.b); e.act = ACT_RELU;
        CK(b.transposed(ops, b.par(L.w), L.cin, L.cout, L.cout * 9, 9, b.src(c3.c_str(), 128, H2, W2), H2, W2, dc.c_str(), e));
    }
    CK(b.conv(ops, layer(pl, d + "conv5.0"), {b.src(dc.c_str(), 64, H, W), b.src(c1.c_str(), 64, H, W)}, H, W, 1, c5.c_str(), ACT_RELU));
    CK(b.conv(ops, layer(pl, d + "conv7.0"), {b.src(c5.c_str(), 64, H, W), b.src(c0.c_str(), 32, H, W)}, H, W, 1, c7.c_str(), ACT_NONE));
    CK(b.conv(ops, layer(pl, d + "recon"), {b.src(c7.c_str(), 64, H, W)}, H, W, 1, RL.c_str(), ACT_SIGMOID));
    return 0;
}

// fused inference tail (tail_kernels.hip) in place of feature_fusion + final_conv + compose
void push_tail(Builder& b, std::vector<Fn>& ops)
{
    Plan& pl = b.pl;
    if (b.dry) return;
    const std::string i = "illum_adjust_net.";
    const LayerP Lu = layer(pl, i + "feature_fusion.0"), Lf = layer(pl, i + "final_conv");
    const float* wf = pl.P + Lu.w; const float* bf = pl.P + Lu.b; const float* wl = pl.P + Lf.w; const float* bl = pl.P + Lf.b;
    float* wc = pl.ws + pl.tailw_off;
    ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_tail_weights(wf, bf, wl, bl, wc, st); }, K_ELEMENTWISE, 0.0, "tail weights"));
    const void* d1 = pl.buf("d1"); const void* d2 = pl.buf("d2"); const void* d3 = pl.buf("d3");
    const float* RL = pl.buf("RL_1"); float* D = pl.buf("D"); float* S = pl.buf("S");
    const int h16 = b.h16, N = pl.N, H = pl.H, W = pl.W, H2 = pl.H2, W2 = pl.W2, H4 = pl.H4, W4 = pl.W4, rl = pl.CRL, cx = pl.CX, B = pl.B;
    const double fl = 2.0 * N * H * W * (192.0 * 64 + 64.0 * 9);      // algorithmic FLOPs of the two layers it replaces
    ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_tail(d1, d2, d3, h16, N, H, W, H2, W2, H4, W4, wc, RL, rl, D, 4, S, cx, B, st); },
                     K_ELEMENTWISE, fl, "fused tail: fusion 1x1 + final 3x3 + compose"));
    {   // reads d1, d2, d3 (storage precision) and R|I (fp32), writes I_delta and S (fp32)
        const double ea = h16 ? 2.0 : 4.0;
        ops.back().bytes = (double)N * (64.0 * ea * ((double)H * W + (double)H2 * W2 + (double)H4 * W4) + (double)H * W * 4.0 * (B + 1 + 1 + B));
    }
}

int build_illum_fwd(Builder& b, std::vector<Fn>& ops, bool fused_tail = false)
{
    Plan& pl = b.pl;
    const int H = pl.H, W = pl.W, H2 = pl.H2, W2 = pl.W2, H4 = pl.H4, W4 = pl.W4, H8 = pl.H8, W8 = pl.W8;
    const std::string i = "illum_adjust_net.";
    CK(b.conv(ops, layer(pl, i + "conv0.0"), {b.src("RL_1", pl.CRL, H, W)}, H, W, 1, "a0", ACT_NONE));
    CK(b.conv(ops, layer(pl, i + "conv1.0"), {b.src("a0", 64, H, W)}, H, W, 2, "a1", ACT_RELU));
    CK(b.conv(ops, layer(pl, i + "conv2.0"), {b.src("a1", 64, H2, W2)}, H2, W2, 2, "a2", ACT_RELU));
    CK(b.conv(ops, layer(pl, i + "conv3.0"), {b.src("a2", 64, H4, W4)}, H4, W4, 2, "a3", ACT_RELU));
    // TransformerBlock (model.py:99-119): tokens = NHWC pixels of a3
    if (b.h16 || !pl.qkv_fused) {
        CK(b.conv(ops, layer(pl, i + "attn.q_linear"), {b.src("a3", 64, H8, W8)}, H8, W8, 1, "qkv", ACT_NONE, nullptr, nullptr, 0));
        CK(b.conv(ops, layer(pl, i + "attn.k_linear"), {b.src("a3", 64, H8, W8)}, H8, W8, 1, "qkv", ACT_NONE, nullptr, nullptr, 64));
        CK(b.conv(ops, layer(pl, i + "attn.v_linear"), {b.src("a3", 64, H8, W8)}, H8, W8, 1, "qkv", ACT_NONE, nullptr, nullptr, 128));
    } else CK(b.qkv_fwd(ops));
    if (!b.dry) {
        const float* qkv = pl.buf("qkv"); float* ao = pl.buf("ao"); float* lse = pl.buf("lse");
        const int N = pl.N, T = H8 * W8;
        if (b.h16) {
            float* aoh = pl.buf("aoh");
            // pre-converted bf16 keys / values go into "gqkv" (a backward-pass tensor: idle during the enhance-only forward)
            float* kvs = pl.buf("gqkv"); const size_t kvs_bytes = (size_t)N * T * 192 * sizeof(float);
            ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_attn_fwd_bf16(qkv, 192, aoh, 64, N, T, st, kvs, kvs_bytes); }, K_ATTN, 4.0 * N * 4 * (double)T * T * 16));
            ops.back().bytes = (double)N * T * (192.0 * 4 + 64.0 * 2);
        } else {
            ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_attn_fwd(qkv, 192, ao, 64, lse, N, T, st); }, K_ATTN, 4.0 * N * 4 * (double)T * T * 16));
            ops.back().bytes = (double)N * T * (192.0 + 64.0 + 4.0) * 4;
        }
    }
    CK(b.conv(ops, layer(pl, i + "attn.ff_linear1"), {b.src("ao", 64, H8, W8)}, H8, W8, 1, "f1", ACT_RELU));
    CK(b.conv(ops, layer(pl, i + "attn.ff_linear2"), {b.src("f1", 64, H8, W8)}, H8, W8, 1, "t3", ACT_NONE, "a3"));
    CK(b.conv(ops, layer(pl, i + "deconv1.0"), {b.src("t3", 64, H4, W4)}, H4, W4, 1, "d1", ACT_RELU, "a2", "u1"));
    CK(b.conv(ops, layer(pl, i + "deconv2.0"), {b.src("d1", 64, H2, W2)}, H2, W2, 1, "d2", ACT_RELU, "a1", "u2"));
    CK(b.conv(ops, layer(pl, i + "deconv3.0"), {b.src("d2", 64, H, W)}, H, W, 1, "d3", ACT_RELU, "a0", "u3"));
    if (fused_tail) { push_tail(b, ops); return 0; }
    CK(b.conv(ops, layer(pl, i + "feature_fusion.0"), {b.src("d1", 64, H, W), b.src("d2", 64, H, W), b.src("d3", 64, H, W)}, H, W, 1, "f", ACT_NONE));
    if (b.h16 || !g_skinny_final) {
        CK(b.conv(ops, layer(pl, i + "final_conv"), {b.src("f", 64, H, W)}, H, W, 1, "D", ACT_NONE));
    } else if (!b.dry) {       // 1-channel output: HBM-bound VALU kernel instead of a 32-wide MFMA tile (tail_kernels.hip)
        const LayerP Lf = layer(pl, i + "final_conv");
        const float* fp = pl.buf("f"); const float* w = pl.P + Lf.w; const float* bi = pl.P + Lf.b; float* D = pl.buf("D");
        const int N = pl.N;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_skinny_fwd(fp, w, bi, D, 4, N, H, W, st); }, K_ELEMENTWISE,
                         2.0 * N * H * W * 64.0 * 9, "final_conv fwd (VALU)"));
        ops.back().bytes = 4.0 * N * H * W * 65.0;
    }
    if (!b.dry) {
        const float* RL = pl.buf("RL_1"); const float* D = pl.buf("D"); float* S = pl.buf("S");
        const int rl = pl.CRL, cx = pl.CX, B = pl.B; const long npix = (long)pl.N * H * W;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_compose(RL, rl, D, 4, S, cx, npix, B, st); }, K_ELEMENTWISE, 0.0, "compose"));
        ops.back().bytes = 4.0 * npix * (2.0 * B + 2.0);
    }
    return 0;
}

// backward of one decomposition pass; G8 holds dL/d(pre-sigmoid recon output)
int build_decomposition_bwd(Builder& b, std::vector<Fn>& ops, const char* xin, int p, bool input_grad)
{
    Plan& pl = b.pl;
    const int H = pl.H, W = pl.W, H2 = pl.H2, W2 = pl.W2;
    auto nm = [&](const char* s) { return std::string(s) + (p == 1 ? "_1" : "_2"); };
    const std::string c0 = nm("c0"), sh = nm("sh"), c1 = nm("c1"), c2 = nm("c2"), c3 = nm("c3"), dc = nm("dc"), c5 = nm("c5"), c7 = nm("c7");
    const std::string d = "decomposition_net.";
    const LayerP Lr = layer(pl, d + "recon"), L7 = layer(pl, d + "conv7.0"), L5 = layer(pl, d + "conv5.0"), Ld = layer(pl, d + "deconv.0", true),
                 L3 = layer(pl, d + "conv3.0"), L2 = layer(pl, d + "conv2.0"), L1 = layer(pl, d + "conv1.0"), Ls = layer(pl, d + "shallow_conv.0"),
                 L0 = layer(pl, d + "conv0.0");
    // gradient buffers of this pass; weight gradients are emitted by the pass-1 builder only, over [pass 1 ; pass 2]
    // (pass 2 ran its data gradients earlier and left its G tensors in the second halves of the pairs)
    auto gn = [&](const char* s) { return p == 1 ? std::string(s) : std::string(s) + "_2"; };
    const std::string G8 = gn("G8"), G7 = gn("G7"), G5 = gn("G5"), G0 = gn("G0"), Gdc = gn("Gdc"), G3 = gn("G3"), G2 = gn("G2"),
                      G1 = gn("G1"), Gsh = gn("Gsh");
    const bool wg = p == 1;
    if (wg) CK(b.wgrad(ops, Lr, 1, b.src("c7_1", 64, H, W), 64, H, W, 0, "G8", 0, true, 2));
    CK(b.dgrad(ops, Lr, 1, G8.c_str(), 0, 0, 64, G7.c_str(), nullptr, 0, 0));
    if (wg) CK(b.wgrad(ops, L7, 1, b.src("c5_1", 64, H, W), 64, H, W, 0, "G7", 0, false, 2));
    if (wg) CK(b.wgrad(ops, L7, 1, b.src("c0_1", 32, H, W), 32, H, W, 64, "G7", 0, true, 2));
    CK(b.dgrad(ops, L7, 1, G7.c_str(), 0, 0, 64, G5.c_str(), c5.c_str(), MASK_RELU, 0));
    CK(b.dgrad(ops, L7, 1, G7.c_str(), 0, 64, 32, G0.c_str(), c0.c_str(), MASK_RELU, 0));
    if (wg) CK(b.wgrad(ops, L5, 1, b.src("dc_1", 64, H, W), 64, H, W, 0, "G5", 0, false, 2));
    if (wg) CK(b.wgrad(ops, L5, 1, b.src("c1_1", 64, H, W), 64, H, W, 64, "G5", 0, true, 2));
    CK(b.dgrad(ops, L5, 1, G5.c_str(), 0, 0, 64, Gdc.c_str(), dc.c_str(), MASK_RELU, 0));
    CK(b.dgrad(ops, L5, 1, G5.c_str(), 0, 64, 64, G1.c_str(), c1.c_str(), MASK_RELU, 0));
    // ConvTranspose2d: wgrad with swapped roles, dgrad = stride-2 conv of Gdc with W read as OIHW (O = ci, I = co)
    if (wg) {
        SrcDesc gs = b.src("Gdc", 64, H, W);
        WgradParams wp; TapList t = ssie_taps_conv(3);
        const BufInfo& cb = pl.bi("c3_1");
        CK(ssie_make_wgrad(wp, gs, pl.N * 2, H, W, 0, b.ptr("c3_1"), cb.cs, 0, 128, H2, W2, 2, t, nullptr, kWgs));
        const size_t need = ssie_wgrad_slab_floats(wp);
        float* slabs = b.take_slabs(need);
        if (!b.dry) {
            wp.slabs = slabs;
            const int sl = pl.slab_seq++ & 1;
            float* dw = pl.G + Ld.w;
            ops.push_back(Fn([wp](hipStream_t st) { return ssie_launch_wgrad(wp, st); }, K_WGRAD, 2.0 * pl.N * 2 * H2 * W2 * 128.0 * 64 * 9, "", sl, SLAB_WRITE));
            ops.back().bytes = 4.0 * ((double)pl.N * 2 * ((double)H * W * 64 + (double)H2 * W2 * 128) + (double)need);
            b.reduce(ops, ssie_make_reduce(slabs, wp.nslices, wp.ntaps, wp.ci_pad, wp.co_pad, 64, 128, dw, 64L * 9, 9, 1, nullptr, nullptr, 1),
                     4.0 * ((double)need + 2.0 * 64 * 128 * 9), sl);
        }
    }
    if (wg) b.bias_grad(ops, Ld, "Gdc", 0, 2);
    {
        TapList t = ssie_taps_conv(3);
        float* wpk = b.take_pack(ssie_packed_floats(64, 128, 9));
        if (!b.dry) {
            pl.packs.push_back(ssie_make_pack(pl.P + Ld.w, wpk, 64, 128, t, 9, 64 * 9, 1));
            SrcDesc in = b.src(Gdc.c_str(), 64, H, W);
            Epilogue e = b.bwd_epi(c3.c_str(), MASK_RELU, 0);
            ConvParams cp; const BufInfo& ob = pl.bi(G3.c_str());
            CK(ssie_make_conv(cp, &in, 1, pl.N, H, W, t, 2, H2, W2, wpk, 128, pl.buf(G3.c_str()), ob.H, ob.W, ob.cs, 0, 1, 0, 0, e));
            b.push(ops, cp, 64);
        }
    }
    if (wg) CK(b.wgrad(ops, L3, 1, b.src("c2_1", 128, H2, W2), 128, H2, W2, 0, "G3", 0, true, 2));
    CK(b.dgrad(ops, L3, 1, G3.c_str(), 0, 0, 128, G2.c_str(), c2.c_str(), MASK_RELU, 0));
    if (wg) CK(b.wgrad(ops, L2, 2, b.src("c1_1", 64, H, W), 64, H, W, 0, "G2", 0, true, 2));
    CK(b.dgrad(ops, L2, 2, G2.c_str(), 0, 0, 64, G1.c_str(), c1.c_str(), MASK_RELU, 1));
    if (wg) CK(b.wgrad(ops, L1, 1, b.src("sh_1", 64, H, W), 64, H, W, 0, "G1", 0, true, 2));
    CK(b.dgrad(ops, L1, 1, G1.c_str(), 0, 0, 64, Gsh.c_str(), nullptr, 0, 0));
    if (wg && pl.spectral) CK(b.spec_wgrad(ops, Ls, "Gsh"));   // (incl. the bias gradient)
    else if (wg) CK(b.wgrad(ops, Ls, 1, b.src("x", pl.CX, H, W), pl.B, H, W, 0, "Gsh", 0, true, 2));
    if (wg) CK(b.wgrad(ops, L0, 1, b.src("x", pl.CX, H, W), pl.B, H, W, 0, "G0", 0, true, 2));
    if (input_grad) {
        if (pl.spectral) b.spec_dgrad(ops, Gsh.c_str(), "gS");
        else CK(b.dgrad(ops, Ls, 1, Gsh.c_str(), 0, 0, pl.B, "gS", nullptr, 0, 1));
        CK(b.dgrad(ops, L0, 1, G0.c_str(), 0, 0, pl.B, "gS", nullptr, 0, 1));
    }
    return 0;
}

int build_illum_bwd(Builder& b, std::vector<Fn>& ops)
{
    Plan& pl = b.pl;
    const int H = pl.H, W = pl.W, H2 = pl.H2, W2 = pl.W2, H4 = pl.H4, W4 = pl.W4, H8 = pl.H8, W8 = pl.W8;
    const std::string i = "illum_adjust_net.";
    const LayerP Lf = layer(pl, i + "final_conv"), Lu = layer(pl, i + "feature_fusion.0"), Ld3 = layer(pl, i + "deconv3.0"),
                 Ld2 = layer(pl, i + "deconv2.0"), Ld1 = layer(pl, i + "deconv1.0"), Lff2 = layer(pl, i + "attn.ff_linear2"),
                 Lff1 = layer(pl, i + "attn.ff_linear1"), Lq = layer(pl, i + "attn.q_linear"), Lk = layer(pl, i + "attn.k_linear"),
                 Lv = layer(pl, i + "attn.v_linear"), Lc3 = layer(pl, i + "conv3.0"), Lc2 = layer(pl, i + "conv2.0"),
                 Lc1 = layer(pl, i + "conv1.0"), Lc0 = layer(pl, i + "conv0.0");
    if (!g_skinny_final) {
        CK(b.wgrad(ops, Lf, 1, b.src("f", 64, H, W), 64, H, W, 0, "gD", 0, true));
        CK(b.dgrad(ops, Lf, 1, "gD", 0, 0, 64, "Gf", nullptr, 0, 0));
    } else if (!b.dry) {
        const float* fp = pl.buf("f"); const float* gD = pl.buf("gD"); float* Gf = pl.buf("Gf"); const float* w = pl.P + Lf.w;
        float* part = pl.ws + pl.skinny_off; float* dw = pl.G + Lf.w; float* db = pl.G + Lf.b;
        const int N = pl.N;
        const double fl = 2.0 * N * H * W * 64.0 * 9;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_skinny_wgrad(fp, gD, 4, part, dw, db, N, H, W, st); }, K_ELEMENTWISE, fl, "final_conv wgrad (VALU)"));
        ops.back().bytes = 4.0 * N * H * W * 65.0;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_skinny_dgrad(gD, 4, w, Gf, N, H, W, st); }, K_ELEMENTWISE, fl, "final_conv dgrad (VALU)"));
        ops.back().bytes = 4.0 * N * H * W * 65.0;
    }
    // feature_fusion is a 1 x 1 convolution over cat[up(d1), up(d2), d3] (model.py:169-173).  A 1 x 1 convolution commutes with
    // nearest up-sampling, so the gradients of the d1 / d2 parts are taken at THEIR resolution from the up-sampling adjoint of Gf
    // (Gf2 = U2^T Gf, Gf4 = U4^T Gf):  dW_k = sum_p Gf[p] (x) up(d_k)[p] = sum_q (U^T Gf)[q] (x) d_k[q]  and  gd_k = U^T (W_k^T Gf) =
    // W_k^T (U^T Gf) - 4x / 16x fewer positions than the full-resolution launches + adjoint of the result (same sums, reassociated).
    b.upadj(ops, "Gf", H, W, "Gf2", 0);
    b.upadj(ops, "Gf", H, W, "Gf4", 0);
    CK(b.wgrad(ops, Lu, 1, b.src("d1", 64, H4, W4), 64, H4, W4, 0, "Gf4"));
    CK(b.wgrad(ops, Lu, 1, b.src("d2", 64, H2, W2), 64, H2, W2, 64, "Gf2"));
    CK(b.wgrad(ops, Lu, 1, b.src("d3", 64, H, W), 64, H, W, 128, "Gf", 0, true));
    // level H: d3 = relu(e3) + a0 - the data gradient w.r.t. d3 goes to a0 as it is (gd3) and to e3 through relu'(u3) (Ge3): both from
    // the one launch (out2_mode 1), instead of a mask launch over the full-resolution tensor
    if (pl.fold_masks) CK(b.dgrad(ops, Lu, 1, "Gf", 0, 128, 64, "Ge3", "u3", MASK_RELU, 0, "gd3", 1));
    else CK(b.dgrad(ops, Lu, 1, "Gf", 0, 128, 64, "gd3", nullptr, 0, 0));
    CK(b.dgrad(ops, Lu, 1, "Gf2", 0, 64, 64, "gd2", nullptr, 0, 0));
    CK(b.dgrad(ops, Lu, 1, "Gf4", 0, 0, 64, "gd1", nullptr, 0, 0));
    if (!pl.fold_masks) b.mask_axpy(ops, "gd3", "u3", MASK_RELU, "Ge3", 64, 0);
    CK(b.wgrad(ops, Ld3, 1, b.src("d2", 64, H, W), 64, H, W, 0, "Ge3", 0, true));
    CK(b.dgrad(ops, Ld3, 1, "Ge3", 0, 0, 64, "tmpH", nullptr, 0, 0));
    // level H/2: d2 = relu(e2) + a1.  tmpH is reused with the (H2, W2) geometry through a view buffer
    if (pl.fold_masks) b.upadj(ops, "tmpH", H, W, "gd2", 1, "u2", "Ge2");
    else { b.upadj(ops, "tmpH", H, W, "gd2", 1); b.mask_axpy(ops, "gd2", "u2", MASK_RELU, "Ge2", 64, 0); }
    CK(b.wgrad(ops, Ld2, 1, b.src("d1", 64, H2, W2), 64, H2, W2, 0, "Ge2", 0, true));
    CK(b.dgrad(ops, Ld2, 1, "Ge2", 0, 0, 64, "tmpH2", nullptr, 0, 0));
    // level H/4
    if (pl.fold_masks) b.upadj(ops, "tmpH2", H2, W2, "gd1", 1, "u1", "Ge1");
    else { b.upadj(ops, "tmpH2", H2, W2, "gd1", 1); b.mask_axpy(ops, "gd1", "u1", MASK_RELU, "Ge1", 64, 0); }
    CK(b.wgrad(ops, Ld1, 1, b.src("t3", 64, H4, W4), 64, H4, W4, 0, "Ge1", 0, true));
    CK(b.dgrad(ops, Ld1, 1, "Ge1", 0, 0, 64, "tmpH4", nullptr, 0, 0)); b.upadj(ops, "tmpH4", H4, W4, "gt3", 0);
    // TransformerBlock backward: t3 = a3 + ff2(relu(ff1(attn(q,k,v(a3)))))
    CK(b.wgrad(ops, Lff2, 1, b.src("f1", 64, H8, W8), 64, H8, W8, 0, "gt3", 0, true));
    CK(b.dgrad(ops, Lff2, 1, "gt3", 0, 0, 64, "gf1", "f1", MASK_RELU, 0));
    CK(b.wgrad(ops, Lff1, 1, b.src("ao", 64, H8, W8), 64, H8, W8, 0, "gf1", 0, true));
    CK(b.dgrad(ops, Lff1, 1, "gf1", 0, 0, 64, "gao", nullptr, 0, 0));
    if (!b.dry) {
        const float* qkv = pl.buf("qkv"); const float* ao = pl.buf("ao"); const float* gao = pl.buf("gao");
        const float* lse = pl.buf("lse"); float* delta = pl.buf("delta"); float* gqkv = pl.buf("gqkv");
        const int N = pl.N, T = H8 * W8;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_attn_bwd(qkv, 192, ao, gao, 64, lse, delta, gqkv, N, T, st); }, K_ATTN, 8.0 * N * 4 * (double)T * T * 16));
    }
    // g(a3) = [gt3 + dgrad_q + dgrad_k + dgrad_v] * relu'(a3)
    if (!pl.qkv_fused) {
        CK(b.wgrad(ops, Lq, 1, b.src("a3", 64, H8, W8), 64, H8, W8, 0, "gqkv", 0, true));
        CK(b.wgrad(ops, Lk, 1, b.src("a3", 64, H8, W8), 64, H8, W8, 0, "gqkv", 64, true));
        CK(b.wgrad(ops, Lv, 1, b.src("a3", 64, H8, W8), 64, H8, W8, 0, "gqkv", 128, true));
        b.mask_axpy(ops, "gt3", "a3", MASK_RELU, "gt3", 64, 0);
        CK(b.dgrad(ops, Lq, 1, "gqkv", 0, 0, 64, "gt3", "a3", MASK_RELU, 1));
        CK(b.dgrad(ops, Lk, 1, "gqkv", 64, 0, 64, "gt3", "a3", MASK_RELU, 1));
        CK(b.dgrad(ops, Lv, 1, "gqkv", 128, 0, 64, "gt3", "a3", MASK_RELU, 1));
    } else {
        // q | k | v as ONE 64 -> 192 layer (model.py:104-106 apply three Linear layers to the same tokens): one weight-gradient launch
        // whose reduction scatters the three 64-row blocks into the three parameter tensors, one data-gradient launch over K = 192
        const long wx = (long)(Lk.w - Lq.w) - 64 * 64, bx = (long)(Lk.b - Lq.b) - 64;
        if ((long)(Lv.w - Lk.w) - 64 * 64 != wx || (long)(Lv.b - Lk.b) - 64 != bx) return SSIE_E_ARG;
        LayerP Lqkv = Lq; Lqkv.cout = 192;
        CK(b.wgrad(ops, Lqkv, 1, b.src("a3", 64, H8, W8), 64, H8, W8, 0, "gqkv", 0, true, 1, 64, wx, bx));
        b.mask_axpy(ops, "gt3", "a3", MASK_RELU, "gt3", 64, 0);
        CK(b.qkv_dgrad(ops));
    }
    CK(b.wgrad(ops, Lc3, 2, b.src("a2", 64, H4, W4), 64, H4, W4, 0, "gt3", 0, true));
    b.mask_axpy(ops, "gd1", "a2", MASK_RELU, "gd1", 64, 0);
    CK(b.dgrad(ops, Lc3, 2, "gt3", 0, 0, 64, "gd1", "a2", MASK_RELU, 1));
    CK(b.wgrad(ops, Lc2, 2, b.src("a1", 64, H2, W2), 64, H2, W2, 0, "gd1", 0, true));
    b.mask_axpy(ops, "gd2", "a1", MASK_RELU, "gd2", 64, 0);
    CK(b.dgrad(ops, Lc2, 2, "gd1", 0, 0, 64, "gd2", "a1", MASK_RELU, 1));
    CK(b.wgrad(ops, Lc1, 2, b.src("a0", 64, H, W), 64, H, W, 0, "gd2", 0, true));
    CK(b.dgrad(ops, Lc1, 2, "gd2", 0, 0, 64, "gd3", nullptr, 0, 1));
    CK(b.wgrad(ops, Lc0, 1, b.src("RL_1", pl.CRL, H, W), pl.B + 1, H, W, 0, "gd3", 0, true));
    // (the sigmoid mask gRL -> G8 that follows stays its own launch: this data gradient runs on the Winograd kernel, whose epilogue has no
    // second-output forms - adding them slowed every Winograd launch by 4 %)
    CK(b.dgrad(ops, Lc0, 1, "gd3", 0, 0, pl.B + 1, "gRL", nullptr, 0, 1));
    return 0;
}

int check_slab_flags(const std::vector<Fn>& ops);

int build_all(Plan& pl, bool dry)
{
    // the op lists are rebuilt (new coefficients, new buffers): a captured step graph holds the old launches
    if (pl.gexec) { hipGraphExecDestroy(pl.gexec); pl.gexec = nullptr; }
    pl.train_calls = 0; pl.graph_failed = false;
    Builder b(pl, dry);
    pl.fwd.clear(); pl.pass2.clear(); pl.lossbwd.clear(); pl.slab_seq = 0;
    CK(build_decomposition_fwd(b, pl.fwd, "x", 1));
    CK(build_illum_fwd(b, pl.fwd));
    CK(build_decomposition_fwd(b, pl.pass2, "S", 2));
    std::vector<Fn>& ops = pl.lossbwd;
    if (!dry) {
        const int N = pl.N, H = pl.H, W = pl.W, B = pl.B;
        LossParams lp; memset(&lp, 0, sizeof(lp));
        lp.x = pl.buf("x"); lp.x_cs = pl.CX; lp.RL = pl.buf("RL_1"); lp.rl_cs = pl.CRL; lp.D = pl.buf("D"); lp.d_cs = 4;
        lp.S = pl.buf("S"); lp.s_cs = pl.CX; lp.E = pl.buf("RL_2"); lp.e_cs = pl.CRL;
        lp.gRL = pl.buf("gRL"); lp.gD = pl.buf("gD"); lp.gS = pl.buf("gS"); lp.G8b = pl.buf("G8_2");
        lp.N = N; lp.H = H; lp.W = W; lp.B = B;
        lp.c_rec = pl.coefs[0]; lp.c_rf = pl.coefs[1]; lp.c_il = pl.coefs[2]; lp.c_id = pl.coefs[3]; lp.c_sp = pl.coefs[5];
        lp.a1 = pl.coefs[6]; lp.a2 = pl.coefs[7];
        const double n = N, c = B, h = H, w = W;
        lp.inv_n0 = (float)(1.0 / (n * c * h * w)); lp.inv_nIx = (float)(1.0 / (n * h * (w - 1))); lp.inv_nIy = (float)(1.0 / (n * (h - 1) * w));
        lp.inv_nRx = (float)(1.0 / (n * c * h * (w - 1))); lp.inv_nRy = (float)(1.0 / (n * c * (h - 1) * w));
        lp.inv_nsp = (float)(1.0 / (n * (c - 1) * h * w));
        lp.partials = pl.ws + pl.lpart_off;
        const int nblk = pl.loss_blocks;
        ops.push_back(Fn([lp, nblk](hipStream_t st) { return ssie_launch_loss_direct(lp, nblk, st); }, K_LOSS));
        ops.back().bytes = 4.0 * N * H * W * (7.0 * B + 5.0);       // reads x, R|I, S, R_enh, D; writes gRL, gS, G8, gD
        FftParams fp; memset(&fp, 0, sizeof(fp));
        fp.x = lp.x; fp.x_cs = lp.x_cs; fp.S = lp.S; fp.s_cs = lp.s_cs; fp.gS = lp.gS; fp.mask = (const uint8_t*)(pl.ws + pl.mask_off);
        fp.N = N; fp.B = B; fp.H = H; fp.W = W; ssie_fft_set_logs(fp);
        fp.ws = pl.ws + pl.fftws_off; fp.ws_floats = pl.fftws_floats; fp.npartials = pl.fft_blocks;
        fp.scale_g = (float)(pl.coefs[4] / (n * c * h * w)); fp.inv_n0 = lp.inv_n0; fp.partials = pl.ws + pl.fpart_off;
        ops.push_back(Fn([fp](hipStream_t st) { return ssie_launch_fft_loss(fp, st); }, K_FFT));
        ops.back().bytes = 4.0 * N * H * W * 4.0 * B;               // reads x and S, read-modify-writes gS
        const float* lpart = pl.ws + pl.lpart_off; const float* fpart = pl.ws + pl.fpart_off; float* scal = pl.ws + pl.scal_off;
        const int nf = pl.fft_blocks;
        float cf[6] = {pl.coefs[0], pl.coefs[1], pl.coefs[2], pl.coefs[3], pl.coefs[4], pl.coefs[5]};
        std::vector<float> cfv(cf, cf + 6);
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_loss_finalize(lpart, nblk, fpart, nf, cfv.data(), scal, st); }, K_LOSS));
    }
    CK(build_decomposition_bwd(b, ops, "S", 2, true));
    if (!dry) {
        const float* gS = pl.buf("gS"); const float* RL = pl.buf("RL_1"); float* gRL = pl.buf("gRL");
        const float* D = pl.buf("D"); float* gD = pl.buf("gD");
        const int cx = pl.CX, rl = pl.CRL, B = pl.B; const long npix = (long)pl.N * pl.H * pl.W;
        ops.push_back(Fn([=](hipStream_t st) { return ssie_launch_product_node(gS, cx, RL, gRL, rl, D, gD, 4, npix, B, st); }, K_ELEMENTWISE, 0.0, "product_node"));
        ops.back().bytes = 4.0 * npix * (B + (B + 1.0) + 2.0 * (B + 1.0) + 1.0 + 2.0);      // gS, R|I, gRL read-modify-write, D, gD read-modify-write
    }
    CK(build_illum_bwd(b, ops));
    b.mask_axpy(ops, "gRL", "RL_1", MASK_SIGMOID, "G8", pl.B + 1, 0);
    CK(build_decomposition_bwd(b, ops, "x", 1, false));
    b.flush_reduces(ops);          // nothing in this list reads a weight gradient: Adam / the all-reduce follow the whole list
    // bf16 inference list (its packs follow the fp32 ones)
    pl.npacks_train = pl.packs.size();
    pl.fwd16.clear();
    const bool tail_ok = pl.fused_tail && ssie_tail_supported(pl.H, pl.W, pl.H2, pl.W2, pl.H4, pl.W4) != 0;
    {   // any band count (model.py:229-234 takes any): bf16 pixels are read in 16-byte (8-channel) slots, so the bf16 input cube
        // "xh" and the bf16 twin "RLh" of the R|I output have pixel strides padded to 8 (zero channels), whatever B is
        b.h16 = true;           // (the bf16 copy of the input cube is written by ingest16(): strided fp32 in, NHWC bf16 out, one pass)
        CK(build_decomposition_fwd(b, pl.fwd16, "x", 1));
        CK(build_illum_fwd(b, pl.fwd16, tail_ok));
        b.h16 = false;
    }
    // fp32 enhance-only list: the training forward minus its last three launches (feature_fusion, final_conv, compose) plus
    // the fused tail; training keeps `fwd` (the two layers' weight gradients need the tensor f)
    pl.fwdi.clear();
    if (!dry && tail_ok && pl.fwd.size() > 3) {
        pl.fwdi.assign(pl.fwd.begin(), pl.fwd.end() - 3);
        push_tail(b, pl.fwdi);
    }
    pl.pack_floats_total = pl.pack_cursor;
    // a debug switch flipped between create and bind can change slice counts: refuse rather than write past the reservation
    if (!dry && pl.pack_cap && pl.pack_cursor > pl.pack_cap) return SSIE_E_WORKSPACE;
    CK(check_slab_flags(pl.lossbwd));
    return 0;
}

int run_ops(std::vector<Fn>& ops, hipStream_t st)
{
    for (auto& f : ops) { int rc = f(st); if (rc) return SSIE_E_LAUNCH; }
    return 0;
}

int g_graph = 0;        // ssie_debug_set_graph: 1 = the train step behind the ingest is replayed as one hipGraph (ssie_plan_loss_fwd_bwd)
int g_fused_tail = 1;   // ssie_debug_set_fused_tail: 0 = inference runs feature_fusion / final_conv / compose as separate launches
int g_overlap = 0;      // ssie_debug_set_overlap: 1 = the weight gradients' slab reductions on a side stream (run_ops_overlapped).  Default 0
                        // since round 3: the persistent convolution kernels leave a side stream no CU to overlap on, and with the wider
                        // reduction kernel launch order on one stream is 0.07 ms per step FASTER at 31 bands (equal at 256)
int g_batched_reduce = 1;   // ssie_debug_set_batched_reduce: 0 = one slab reduction per layer, right behind its weight-gradient launch (plans built afterwards)

// backward schedule with the slab reductions on the side stream:
//   wgrad(slab b)   on st   : waits for the reduction that last read slab b, then records ev_w[b]
//   reduce(slab b)  on side : waits for ev_w[b], records ev_r[b]
// and st joins the side stream at the end, so everything after the call (Adam, the gradient all-reduce) sees the
// complete gradient buffer.  Reductions stay in launch order among themselves (one side stream), which keeps the
// two-pass accumulation into the shared decomposition-net gradients deterministic.
int run_ops_overlapped(Plan& pl, std::vector<Fn>& ops, hipStream_t st)
{
    if (!g_overlap) return run_ops(ops, st);
    if (!pl.side) {
        if (hipStreamCreateWithFlags(&pl.side, hipStreamNonBlocking) != hipSuccess) return SSIE_E_LAUNCH;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&pl.ev_w[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&pl.ev_r[i], hipEventDisableTiming) != hipSuccess) return SSIE_E_LAUNCH;
    }
    // the stream order is derived from each op's slab_use flag alone; every event call is checked - a failed record / wait
    // would silently drop an ordering edge, i.e. turn into wrong gradients instead of an error
    bool pending[2] = {false, false};
    int rc = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess) rc = SSIE_E_LAUNCH; return e == hipSuccess; };
    for (auto& f : ops) {
        if (f.slab_use == SLAB_WRITE) {
            if (pending[f.slab]) { if (!ok(hipStreamWaitEvent(st, pl.ev_r[f.slab], 0))) break; pending[f.slab] = false; }
            if (f(st)) { rc = SSIE_E_LAUNCH; break; }
            if (!ok(hipEventRecord(pl.ev_w[f.slab], st))) break;
        } else if (f.slab_use == SLAB_READ && f.slab == 2) {
            // a batched reduction (list built for the other executor): every producer ran on st, so launch order is the dependency
            for (int i = 0; i < 2; ++i) if (pending[i]) { if (!ok(hipStreamWaitEvent(st, pl.ev_r[i], 0))) break; pending[i] = false; }
            if (rc || f(st)) { rc = SSIE_E_LAUNCH; break; }
        } else if (f.slab_use == SLAB_READ) {
            if (!ok(hipStreamWaitEvent(pl.side, pl.ev_w[f.slab], 0))) break;
            if (f(pl.side)) { rc = SSIE_E_LAUNCH; break; }
            pending[f.slab] = true;               // set before the record: the join below must cover this launch either way
            if (!ok(hipEventRecord(pl.ev_r[f.slab], pl.side))) break;
        } else if (f(st)) { rc = SSIE_E_LAUNCH; break; }
    }
    if (rc) {
        // error path: the event chain may be incomplete, so join the side stream the blunt way before reporting
        hipStreamSynchronize(pl.side);
        return rc;
    }
    for (int i = 0; i < 2; ++i) if (pending[i] && hipStreamWaitEvent(st, pl.ev_r[i], 0) != hipSuccess) rc = SSIE_E_LAUNCH;
    return rc;
}

// every op that touches a slab area must say so: a weight-gradient kind without SLAB_WRITE or a reduction kind without
// SLAB_READ would race the side stream (this round's ADVICE: the round-2 failure was exactly a kind missing from a list)
int check_slab_flags(const std::vector<Fn>& ops)
{
    for (auto& f : ops) {
        const bool w = f.kind == K_WGRAD || f.kind == K_WGRAD_WINO, r = f.kind == K_WGRAD_REDUCE;
        if ((w && f.slab_use != SLAB_WRITE) || (r && f.slab_use != SLAB_READ) || (!w && !r && f.slab_use != SLAB_NONE)) return SSIE_E_ARG;
        if (f.slab < 0 || f.slab > 2 || (f.slab == 2 && f.slab_use != SLAB_READ)) return SSIE_E_ARG;
    }
    return 0;
}

} // namespace

// -------------------------------------------------------------------------------------------------
// C-ABI
// -------------------------------------------------------------------------------------------------
extern "C" void* ssie_plan_create(int N, int bands, int H, int W, const float* coefs8)
{
    if (N < 1 || bands < 2 || H < 8 || W < 8 || (H & 1) || (W & 1)) return nullptr;   // model.py:59 needs even H, W
    Plan* pl = new Plan();
    pl->N = N; pl->B = bands; pl->H = H; pl->W = W; pl->fused_tail = g_fused_tail != 0; pl->qkv_fused = g_qkv_fused != 0; pl->fold_masks = g_fold_masks != 0;
    pl->CX = ssie_round_up(bands, 4); pl->CRL = ssie_round_up(bands + 1, 4);
    pl->H2 = (H + 1) / 2; pl->W2 = (W + 1) / 2; pl->H4 = (pl->H2 + 1) / 2; pl->W4 = (pl->W2 + 1) / 2;
    pl->H8 = (pl->H4 + 1) / 2; pl->W8 = (pl->W4 + 1) / 2;
    for (int i = 0; i < 8; ++i) pl->coefs[i] = coefs8 ? coefs8[i] : 0.f;
    build_params(*pl);
    build_buffers(*pl);
    // the kernels address activations with 32-bit element offsets, and the weight gradients run over [pass 1 ; pass 2] pairs of 2N
    // patches: no plan whose largest tensor reaches 2^31 floats (every BASELINE config stays below 2^29)
    for (auto& kv : pl->bufs)
        if (!ssie_fits_i32(2L * kv.second.N, kv.second.H, kv.second.W, kv.second.cs)) { delete pl; return nullptr; }
    // aliases of tmpH with the lower-resolution geometries
    BufInfo t = pl->bufs["tmpH"];
    BufInfo t2 = t; t2.H = pl->H2; t2.W = pl->W2; pl->bufs["tmpH2"] = t2;
    BufInfo t4 = t; t4.H = pl->H4; t4.W = pl->W4; pl->bufs["tmpH4"] = t4;
    if (build_all(*pl, true)) { delete pl; return nullptr; }
    pl->pack_cap = pl->pack_floats_total;
    pl->ws_floats = align_up(pl->pack_off + pl->pack_floats_total + 64, 64);
    return pl;
}

extern "C" void ssie_debug_set_overlap(int on) { g_overlap = on; }
extern "C" void ssie_debug_set_batched_reduce(int on) { g_batched_reduce = on; }
extern "C" void ssie_debug_set_graph(int on) { g_graph = on; }
// product API (include/ssie_hip.h): replay this plan's train step as one hipGraph from its second call on (the first call runs every
// launcher's one-time hipFuncSetAttribute; capture and the first replay happen at the second)
extern "C" int ssie_plan_set_graph(void* h, int on)
{
    Plan* pl = (Plan*)h;
    if (!pl) return SSIE_E_ARG;
    pl->use_graph = on != 0; pl->graph_failed = false;
    if (!on && pl->gexec) { hipGraphExecDestroy(pl->gexec); pl->gexec = nullptr; }
    return 0;
}
extern "C" void ssie_debug_set_spectral9(int on) { g_spectral9 = on; }
extern "C" void ssie_debug_set_qkv_fused(int on) { g_qkv_fused = on; }
extern "C" void ssie_debug_set_fold_masks(int on) { g_fold_masks = on; }
extern "C" void ssie_debug_set_skinny_final(int on) { g_skinny_final = on; }   // takes effect for plans created afterwards
extern "C" void ssie_debug_set_fused_tail(int on) { g_fused_tail = on; }     // takes effect for plans bound afterwards
extern "C" void ssie_plan_destroy(void* h) { delete (Plan*)h; }
extern "C" size_t ssie_plan_workspace_bytes(void* h) { return h ? ((Plan*)h)->ws_floats * 4 : 0; }
extern "C" size_t ssie_plan_param_floats(void* h) { return h ? ((Plan*)h)->nparam_floats : 0; }
extern "C" int ssie_plan_num_params(void* h) { return h ? (int)((Plan*)h)->params.size() : 0; }

extern "C" int ssie_plan_param_info(void* h, int idx, char* name, int name_cap, size_t* off, int* ndim, int* shape4)
{
    Plan* pl = (Plan*)h;
    if (!pl || idx < 0 || idx >= (int)pl->params.size()) return SSIE_E_ARG;
    const ParamInfo& p = pl->params[idx];
    if (name) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (off) *off = p.off; if (ndim) *ndim = p.ndim;
    if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = p.shape[i];
    return 0;
}

extern "C" int ssie_plan_buffer(void* h, const char* name, size_t* off_floats, int* dims5)
{
    Plan* pl = (Plan*)h;
    if (!pl || !name) return SSIE_E_ARG;
    if (!strcmp(name, "scalars")) { *off_floats = pl->scal_off; dims5[0] = 1; dims5[1] = 1; dims5[2] = 1; dims5[3] = 7; dims5[4] = 16; return 0; }
    auto it = pl->bufs.find(name);
    if (it == pl->bufs.end()) return SSIE_E_ARG;
    *off_floats = it->second.off;
    dims5[0] = it->second.N; dims5[1] = it->second.H; dims5[2] = it->second.W; dims5[3] = it->second.C; dims5[4] = it->second.cs;
    return 0;
}

extern "C" int ssie_plan_set_coefs(void* h, const float* coefs8)
{
    Plan* pl = (Plan*)h;
    if (!pl || !coefs8) return SSIE_E_ARG;
    for (int i = 0; i < 8; ++i) pl->coefs[i] = coefs8[i];
    if (pl->bound) return build_all(*pl, false);
    return 0;
}

extern "C" int ssie_plan_bind(void* h, void* workspace, size_t ws_bytes, float* params, float* grads, void* stream)
{
    Plan* pl = (Plan*)h;
    if (!pl || !workspace || !params) return SSIE_E_ARG;
    if (ws_bytes < pl->ws_floats * 4) return SSIE_E_WORKSPACE;
    if (((uintptr_t)workspace) % 256) return SSIE_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    pl->ws = (float*)workspace; pl->P = params; pl->G = grads;
    if (hipMemsetAsync(workspace, 0, pl->ws_floats * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
    int rc = build_all(*pl, false);
    if (rc) return rc;
    if (pl->packs.size() > 384) return SSIE_E_WORKSPACE;
    if (hipMemcpyAsync(pl->ws + pl->packdesc_off, pl->packs.data(), pl->packs.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st) != hipSuccess) return SSIE_E_LAUNCH;
    if (ssie_fft_supported(pl->H, pl->W)) {
        std::vector<uint8_t> m((size_t)pl->H * pl->W);
        ssie_fourier_mask_host(pl->H, pl->W, 0.1f, m.data());
        if (hipMemcpyAsync(pl->ws + pl->mask_off, m.data(), m.size(), hipMemcpyHostToDevice, st) != hipSuccess) return SSIE_E_LAUNCH;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return SSIE_E_LAUNCH;     // host staging buffers go out of scope
    pl->bound = true;
    return 0;
}

static int pack_all(Plan* pl, hipStream_t st)
{
    // tile-queue counters of every conv launch of this step (one memset node)
    if (hipMemsetAsync(pl->ws + pl->counter_off, 0, 1024 * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
    return ssie_launch_pack_batched((const PackDesc*)(pl->ws + pl->packdesc_off), (int)pl->npacks_train, st) ? SSIE_E_LAUNCH : 0;
}

static int pack_all_bf16(Plan* pl, hipStream_t st)
{
    if (hipMemsetAsync(pl->ws + pl->counter_off, 0, 1024 * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
    const int n16 = (int)(pl->packs.size() - pl->npacks_train);
    return ssie_launch_pack_batched((const PackDesc*)(pl->ws + pl->packdesc_off) + pl->npacks_train, n16, st) ? SSIE_E_LAUNCH : 0;
}

static int ingest(Plan* pl, const float* x, const long* strides4, hipStream_t st)
{
    return ssie_launch_ingest(x, strides4[0], strides4[1], strides4[2], strides4[3], pl->buf("x"), pl->N, pl->B, pl->H, pl->W, pl->CX, st)
               ? SSIE_E_LAUNCH : 0;
}

// bf16 enhance-only path: the input goes straight into the bf16 NHWC buffer "xh" (the fp32 copy "x" is not written)
static int ingest16(Plan* pl, const float* x, const long* strides4, hipStream_t st)
{
    return ssie_launch_ingest_bf16(x, strides4[0], strides4[1], strides4[2], strides4[3], pl->buf("xh"), pl->N, pl->B, pl->H, pl->W,
                                   ssie_round_up(pl->CX, 8), st) ? SSIE_E_LAUNCH : 0;
}

extern "C" int ssie_plan_enhance_fwd(void* h, const float* x, const long* strides4, void* stream)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    CK(pack_all(pl, st));
    CK(ingest(pl, x, strides4, st));
    return run_ops(pl->fwdi.empty() ? pl->fwd : pl->fwdi, st);
}

// enhance-only forward with bf16 activations / weights and bf16 MFMA (fp32 accumulate, fp32 bias / activation / attention);
// the four outputs (RL_1 = R|I, D, S) are fp32 like ssie_plan_enhance_fwd's
extern "C" int ssie_plan_enhance_fwd_bf16(void* h, const float* x, const long* strides4, void* stream)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4) return SSIE_E_ARG;
    if (pl->fwd16.empty()) return SSIE_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    CK(pack_all_bf16(pl, st));
    CK(ingest16(pl, x, strides4, st));
    return run_ops(pl->fwd16, st);
}

extern "C" int ssie_plan_loss_fwd_bwd(void* h, const float* x, const long* strides4, int with_backward, void* stream)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4) return SSIE_E_ARG;
    if (!ssie_fft_supported(pl->H, pl->W)) return SSIE_E_SHAPE;
    if (with_backward && !pl->G) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (with_backward && (g_graph || pl->use_graph) && !g_overlap && !pl->graph_failed) {
        // One hipGraph for everything behind the ingest (weight packing, both forward passes, losses, backward): the op lists are
        // fixed per plan and every pointer in them belongs to the plan, so the step is captured ONCE (at the second call: the first
        // one has run every launcher's one-time hipFuncSetAttribute) on a stream of the plan's own - the caller's may be the null
        // stream, which cannot be captured - and replayed on the caller's stream.
        CK(ingest(pl, x, strides4, st));
        if (!pl->gexec && pl->train_calls >= 1) {
            hipGraph_t graph = nullptr;
            bool ok = pl->cap_stream || hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                int rc = pack_all(pl, pl->cap_stream);
                if (!rc) rc = run_ops(pl->fwd, pl->cap_stream);
                if (!rc) rc = run_ops(pl->pass2, pl->cap_stream);
                if (!rc && hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, pl->cap_stream) != hipSuccess) rc = SSIE_E_LAUNCH;
                if (!rc) rc = run_ops(pl->lossbwd, pl->cap_stream);
                const bool ended = hipStreamEndCapture(pl->cap_stream, &graph) == hipSuccess;
                ok = !rc && ended && graph && hipGraphInstantiate(&pl->gexec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) hipGraphDestroy(graph);
            }
            if (!ok) { pl->graph_failed = true; pl->gexec = nullptr; (void)hipGetLastError(); }
        }
        if (pl->gexec) {
            ++pl->train_calls;
            if (hipGraphLaunch(pl->gexec, st) == hipSuccess) return 0;
            // a failed replay enqueued nothing: drop the graph for good and run this step's launches eagerly below
            (void)hipGetLastError();
            hipGraphExecDestroy(pl->gexec); pl->gexec = nullptr; pl->graph_failed = true;
        } else ++pl->train_calls;
        CK(pack_all(pl, st));
        CK(run_ops(pl->fwd, st));
        CK(run_ops(pl->pass2, st));
        if (hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
        return run_ops(pl->lossbwd, st);
    }
    CK(pack_all(pl, st));
    CK(ingest(pl, x, strides4, st));
    CK(run_ops(pl->fwd, st));
    CK(run_ops(pl->pass2, st));
    if (with_backward) {
        if (hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
        return run_ops_overlapped(*pl, pl->lossbwd, st);
    }
    // loss only: the first three ops of lossbwd are loss_direct, fft_loss, finalize (they also write cotangents)
    for (int i = 0; i < 3; ++i) { int rc = pl->lossbwd[i](st); if (rc) return SSIE_E_LAUNCH; }
    return 0;
}

// TEST ENTRY (include/ssie_debug.h): the backward schedule alone, on cotangents the caller wrote into the plan buffers
// "gRL" (dL/dR_low | dL/dI_low), "gD", "gS" (direct terms) and "G8_2" (dL/d pre-sigmoid output of pass 2) after a
// ssie_plan_loss_fwd_bwd call filled the activations.  With the sg()-carrying loss kernels out of the way the chain is
// linear in the cotangents, so every parameter gradient can be pinned at a FIXED tolerance.
extern "C" int ssie_plan_backward_from_cotangents(void* h, void* stream)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !pl->G) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
    if (hipMemsetAsync(pl->ws + pl->counter_off, 0, 1024 * 4, st) != hipSuccess) return SSIE_E_LAUNCH;
    std::vector<Fn> tail(pl->lossbwd.begin() + 3, pl->lossbwd.end());
    return run_ops_overlapped(*pl, tail, st);
}

// one full compute_loss + backward with a HIP event after every launch: per-kernel-class device time and
// algorithmic FLOPs of that step (synchronises).  ms/flops/counts have K_NKINDS = SSIE_NKINDS entries (include/ssie_hip.h);
// a Winograd launch is counted with the FLOPs of the direct convolution it replaces (it executes 16/36 of them)
extern "C" int ssie_plan_profile_step(void* h, const float* x, const long* strides4, void* stream,
                                      double* ms, double* flops, int* counts)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4 || !pl->G || !ms || !flops || !counts) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    std::vector<const Fn*> seq;
    for (auto& f : pl->fwd) seq.push_back(&f);
    for (auto& f : pl->pass2) seq.push_back(&f);
    for (auto& f : pl->lossbwd) seq.push_back(&f);
    std::vector<hipEvent_t> ev(seq.size() + 2);
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return SSIE_E_LAUNCH;
    for (int k = 0; k < K_NKINDS; ++k) { ms[k] = 0; flops[k] = 0; counts[k] = 0; }
    hipEventRecord(ev[0], st);
    int rc = pack_all(pl, st);
    hipEventRecord(ev[1], st);
    if (!rc) rc = ingest(pl, x, strides4, st);
    if (!rc && hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, st) != hipSuccess) rc = SSIE_E_LAUNCH;
    hipEventRecord(ev[1], st);       // pack only is attributed below; ingest + memset are folded into the first op
    size_t done = 0;
    for (; !rc && done < seq.size(); ++done) {
        if ((*seq[done])(st)) rc = SSIE_E_LAUNCH;
        hipEventRecord(ev[done + 2], st);
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = SSIE_E_LAUNCH;
    if (!rc) {
        float t = 0.f;
        hipEventElapsedTime(&t, ev[0], ev[1]); ms[K_PACK] += t; counts[K_PACK] += 1;
        for (size_t i = 0; i < seq.size(); ++i) {
            hipEventElapsedTime(&t, ev[i + 1], ev[i + 2]);
            ms[seq[i]->kind] += t; flops[seq[i]->kind] += seq[i]->flops; counts[seq[i]->kind] += 1;
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

// per-launch variant of ssie_plan_profile_step (dev tool): ms/flops/kind per op in launch order + a '\n'-joined tag list
extern "C" int ssie_plan_profile_ops(void* h, const float* x, const long* strides4, void* stream,
                                     double* ms, double* flops, int* kinds, int cap, char* tags, int tags_cap)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4 || !pl->G) return -SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    std::vector<const Fn*> seq;
    for (auto& f : pl->fwd) seq.push_back(&f);
    for (auto& f : pl->pass2) seq.push_back(&f);
    for (auto& f : pl->lossbwd) seq.push_back(&f);
    if ((int)seq.size() > cap) return -SSIE_E_WORKSPACE;
    std::vector<hipEvent_t> ev(seq.size() + 1);
    for (auto& e : ev) hipEventCreate(&e);
    pack_all(pl, st); ingest(pl, x, strides4, st);
    hipMemsetAsync(pl->G, 0, pl->nparam_floats * 4, st);
    hipEventRecord(ev[0], st);
    for (size_t i = 0; i < seq.size(); ++i) { (*seq[i])(st); hipEventRecord(ev[i + 1], st); }
    hipStreamSynchronize(st);
    std::string all;
    for (size_t i = 0; i < seq.size(); ++i) {
        float t = 0.f; hipEventElapsedTime(&t, ev[i], ev[i + 1]);
        ms[i] = t; flops[i] = seq[i]->flops; kinds[i] = seq[i]->kind;
        all += seq[i]->tag; all += "\n";
    }
    for (auto& e : ev) hipEventDestroy(e);
    if (tags && tags_cap > 0) { strncpy(tags, all.c_str(), tags_cap - 1); tags[tags_cap - 1] = 0; }
    return (int)seq.size();
}

// per-launch timing of ONE op list (dev tool): which = 0 enhance forward (fp32), 1 = bf16 enhance forward
extern "C" int ssie_plan_profile_list(void* h, const float* x, const long* strides4, void* stream, int which,
                                      double* ms, double* flops, int* kinds, int cap, char* tags, int tags_cap)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !x || !strides4) return -SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    std::vector<Fn>& seq = which == 1 ? pl->fwd16 : (pl->fwdi.empty() ? pl->fwd : pl->fwdi);
    if ((int)seq.size() > cap) return -SSIE_E_WORKSPACE;
    std::vector<hipEvent_t> ev(seq.size() + 1);
    for (auto& e : ev) hipEventCreate(&e);
    if (which == 1) pack_all_bf16(pl, st); else pack_all(pl, st);
    if (which == 1) ingest16(pl, x, strides4, st); else ingest(pl, x, strides4, st);
    hipEventRecord(ev[0], st);
    for (size_t i = 0; i < seq.size(); ++i) { seq[i](st); hipEventRecord(ev[i + 1], st); }
    hipStreamSynchronize(st);
    std::string all;
    for (size_t i = 0; i < seq.size(); ++i) {
        float t = 0.f; hipEventElapsedTime(&t, ev[i], ev[i + 1]);
        ms[i] = t; flops[i] = seq[i].flops; kinds[i] = seq[i].kind;
        all += seq[i].tag; all += "\n";
    }
    for (auto& e : ev) hipEventDestroy(e);
    if (tags && tags_cap > 0) { strncpy(tags, all.c_str(), tags_cap - 1); tags[tags_cap - 1] = 0; }
    return (int)seq.size();
}

// algorithmic HBM bytes per launch (each operand read once, each result written once; 0 where a launch states none), in the
// order of the matching profile call: which = 0 / 1 = the lists of ssie_plan_profile_list, 2 = the train step of ssie_plan_profile_ops
extern "C" int ssie_plan_op_bytes(void* h, int which, double* bytes, int cap)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !bytes) return -SSIE_E_ARG;
    std::vector<const Fn*> seq;
    if (which == 2) {
        for (auto& f : pl->fwd) seq.push_back(&f);
        for (auto& f : pl->pass2) seq.push_back(&f);
        for (auto& f : pl->lossbwd) seq.push_back(&f);
    } else for (auto& f : (which == 1 ? pl->fwd16 : (pl->fwdi.empty() ? pl->fwd : pl->fwdi))) seq.push_back(&f);
    if ((int)seq.size() > cap) return -SSIE_E_WORKSPACE;
    for (size_t i = 0; i < seq.size(); ++i) bytes[i] = seq[i]->bytes;
    return (int)seq.size();
}

// the same per kernel class of the train step (indices as ssie_plan_profile_step)
extern "C" int ssie_plan_class_bytes(void* h, double* bytes)
{
    Plan* pl = (Plan*)h;
    if (!pl || !pl->bound || !bytes) return SSIE_E_ARG;
    for (int k = 0; k < K_NKINDS; ++k) bytes[k] = 0.0;
    for (auto* l : {&pl->fwd, &pl->pass2, &pl->lossbwd}) for (auto& f : *l) bytes[f.kind] += f.bytes;
    return 0;
}

// number of launches in the three op lists (enhance forward, second decomposition pass, loss + backward)
extern "C" int ssie_plan_num_ops(void* h, int* counts3)
{
    Plan* pl = (Plan*)h;
    if (!pl || !counts3) return SSIE_E_ARG;
    counts3[0] = (int)pl->fwd.size(); counts3[1] = (int)pl->pass2.size(); counts3[2] = (int)pl->lossbwd.size();
    return 0;
}

extern "C" int ssie_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                              float grad_scale, float lr, int step, float beta1, float beta2, float eps, void* stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return SSIE_E_ARG;
    return ssie_launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, grad_scale, lr, step, beta1, beta2, eps, (hipStream_t)stream)
               ? SSIE_E_LAUNCH : 0;
}
