// Host-side walk of libafigan_hip's entry points that need no GPU, for the AddressSanitizer / UndefinedBehaviorSanitizer build
// (afigan_amd/csrc/Makefile: sanitize-check; tests/test_cabi.py runs it).  Test infrastructure: it only calls the C-ABI of include/afigan_hip.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "afigan_hip.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host sanitizer driver: FAILED %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    CHECK(afi_abi_version() == 8);
    CHECK(std::strlen(afi_build_id()) > 0);
    for (int s = -1; s < 8; ++s) CHECK(afi_status_string(s) != nullptr);
    // workspace layouts over a sweep of shapes (small maps, the Winograd thresholds, the benchmark's five levels, odd sizes)
    const int shapes[][3] = {{1, 5, 7}, {2, 7, 11}, {1, 25, 34}, {2, 13, 21}, {1, 32, 32}, {2, 26, 42}, {2, 52, 84}, {2, 104, 168}, {2, 200, 336}, {16, 25, 34}, {1, 33, 47}, {3, 1, 1}};
    for (const auto& sh : shapes) {
        const int N = sh[0], H = sh[1], W = sh[2];
        for (int C : {16, 32, 128, 256}) {
            const int G = C / 8 < 4 ? 4 : C / 8;
            for (int R : {1, 3}) {
                const long long f = afi_generator_fwd_ws_floats(C, G, R, N, H, W), b = afi_generator_bwd_ws_floats(C, G, R, N, H, W);
                CHECK(f > 0 && b > 0);
            }
        }
        for (int F0 : {16, 256}) {
            const int F[4] = {F0, 2 * F0, 4 * F0, 4 * F0};
            long long off[12];
            CHECK(afi_discriminator_fwd_ws_floats(F, N, H, W) > 0 && afi_discriminator_bwd_ws_floats(F, N, H, W) > 0);
            CHECK(afi_discriminator_ws_layout(F, N, H, W, off) == 0);
            for (int i = 1; i < 6; ++i) CHECK(off[i] >= 0);
            CHECK(afi_discriminator_saved_activations(nullptr, F, N, H, W) == 3);        // (y[2] is never written: AFI_OPT_D_FUSE_TAIL)
            // the context-aware size query (ABI v7): never above the context-free upper bound; forwards no backward follows keep no planes; the default
            // context's training forward keeps the planes of blocks 1 and 2 (AFI_OPT_WINOGRAD_F4_FORWARD = 12) where the Winograd F(4x4) forward runs
            const long long all = afi_discriminator_fwd_ws_floats(F, N, H, W);
            const long long t1 = afi_discriminator_fwd_ws_floats_ex(nullptr, F, N, H, W, 1), t2 = afi_discriminator_fwd_ws_floats_ex(nullptr, F, N, H, W, 2);
            CHECK(t2 > 0 && t2 <= t1 && t1 <= all);
            const bool big = (long long)N * H * W >= 8192 && F0 == 256;
            CHECK(big ? (t1 > t2 && t1 <= all) : (t1 == t2));
        }
        CHECK(afi_conv3x3_wino_ws_floats(N, H, W, 256, 512) > 0);
    }
    const int Fbad[4] = {256, 512, 1024, 1024};
    CHECK(afi_discriminator_ws_layout(Fbad, 0, 3, 3, nullptr) != 0);
    // options and the context API without a device: NULL handling, defaults, refusals
    for (int o = -2; o < AFI_OPT_COUNT + 2; ++o) {
        const long long v = afi_ctx_get_option(nullptr, o);
        CHECK((o >= 0 && o < AFI_OPT_COUNT) ? v >= 0 : v == -1);
        CHECK(afi_ctx_set_option(nullptr, o, 1) != 0);
    }
    CHECK(afi_ctx_create(nullptr) != 0 && afi_ctx_destroy(nullptr) == 0);
    CHECK(afi_ctx_set_op_scratch(nullptr, nullptr, 0) != 0 && afi_ctx_set_wino_weight_cache(nullptr, nullptr, 0) != 0);
    CHECK(afi_ctx_set_wino_wgrad_accum(nullptr, nullptr, 0) != 0 && afi_ctx_wino_wgrad_flush(nullptr, nullptr) != 0);
    CHECK(afi_ctx_wino_weight_cache_invalidate(nullptr) != 0 && afi_ctx_wino_wgrad_discard(nullptr) != 0);
    CHECK(afi_ctx_get_compute_dtype(nullptr) == AFI_DTYPE_DEFAULT && afi_ctx_set_compute_dtype(nullptr, AFI_DTYPE_F32) != 0);
    afi_ctx_t* cx = nullptr;
    if (afi_ctx_create(&cx) == 0 && cx) {                     // (a box with a device: the context's host-side state as well)
        for (int o = 0; o < AFI_OPT_COUNT; ++o) { const long long v = afi_ctx_get_option(cx, o); CHECK(afi_ctx_set_option(cx, o, v) == 0 && afi_ctx_get_option(cx, o) == v); }
        for (int d : {AFI_DTYPE_F32, AFI_DTYPE_BF16, AFI_DTYPE_F16X3, AFI_DTYPE_BF16X3, AFI_DTYPE_BF16X6}) CHECK(afi_ctx_set_compute_dtype(cx, d) == 0 && afi_ctx_get_compute_dtype(cx) == d);
        CHECK(afi_ctx_set_compute_dtype(cx, 5) != 0);
        CHECK(afi_ctx_destroy(cx) == 0);
    }
    // scratch-size queries of the stand-alone GEMMs
    for (int d : {AFI_DTYPE_F32, AFI_DTYPE_BF16, AFI_DTYPE_F16X3, AFI_DTYPE_BF16X3, AFI_DTYPE_BF16X6, 5}) {
        const long long a = afi_gemm_nt_scratch_bytes(36, 1024, 1024, d), b = afi_gemm_tn_scratch_bytes(36, d);
        CHECK(d == 5 ? (a == -1 && b == -1) : (a >= 0 && b >= 0));
    }
    CHECK(afi_gemm_nt_scratch_bytes(0, 128, 32, AFI_DTYPE_F16X3) == -1);
    CHECK(afi_gemm_nt(nullptr, nullptr, nullptr, 1, 128, 128, 32, AFI_DTYPE_F16X3, nullptr, 0, nullptr) != 0);
    CHECK(afi_gemm_tn(nullptr, nullptr, nullptr, 1, 128, 128, 128, AFI_DTYPE_F16X3, nullptr, 0, nullptr) != 0);
    // the stream-K plan of the grouped small-map weight gradients: the kernel's own walk, on the host
    unsigned seed = 12345u;
    auto rnd = [&](int lo, int hi) { seed = seed * 1664525u + 1013904223u; return lo + (int)((seed >> 8) % (unsigned)(hi - lo + 1)); };
    for (int it = 0; it < 400; ++it) {
        const int n = rnd(1, 40);
        std::vector<long long> px(n); std::vector<int> tl(n);
        int T = 0;
        for (int i = 0; i < n; ++i) { px[i] = rnd(1, 5000); tl[i] = rnd(1, 40); T += tl[i]; }
        std::vector<int> stored(T), added(T), stages(T);
        CHECK(afi_debug_wgrad_sk_plan(px.data(), tl.data(), n, rnd(1, 4), stored.data(), added.data(), stages.data()) == 0);
        int t = 0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < tl[i]; ++k, ++t) {
                CHECK((stored[t] == 1 && added[t] == 0) || (stored[t] == 0 && added[t] >= 2));
                CHECK(stages[t] == (int)((px[i] + 31) / 32));
            }
    }
    CHECK(afi_profile_num_kinds() > 0);
    for (int k = -1; k <= afi_profile_num_kinds(); ++k) CHECK(afi_profile_kind_name(k) != nullptr);
    std::printf("host sanitizer driver: ok\n");
    return 0;
}
