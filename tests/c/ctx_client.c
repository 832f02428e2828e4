/* ctx_client.c - a plain C99 client of the context API of libss_hip.so (include/ss_hip.h), the way a C / cgo / JNI host
 * of the reference's audio path would use it.  TEST INFRASTRUCTURE.
 *
 *   ctx_client plan                          host only (no GPU): plans two steps, prints the unit descriptors
 *   ctx_client sims                          host only: simulator state columns -> unit columns (ss_ctx_sims_units)
 *   ctx_client observe in.bin out.bin        GPU: reads {sr, n_src, src_len[], src..., R, cap, rir_len[], rir[R][2][cap],
 *                                            n, sound[], t0[], rir_idx[]} and writes audiogoal [n][2][sr] + spectrogram
 *                                            [n][65][T4][2]; device memory through the HIP C API.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ss_hip.h"

#ifdef WITH_HIP
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#endif

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d\n", #x, rc_); return 10; } } while (0)

static int plan_only(void) {
    ss_ctx* ctx = NULL;
    CHECK(ss_ctx_create(&ctx, 16000, 16000, SS_PAD_REFLECT, 0, 8));
    if (ss_ctx_add_source_len(ctx, 16000) != 0 || ss_ctx_add_source_len(ctx, 80000) != 1) return 11;
    CHECK(ss_ctx_set_rir_bank(ctx, NULL, NULL, 32000, 16000, 1, 16000));
    {
        int sound[3] = {0, 1, 1}, t0[3] = {0, 32000, 0}, rir[3] = {5, 6, -1};
        int dis_sound[3] = {0, 0, 0}, dis_rir[3] = {-1, 9, -1};
        ss_units u;
        int desc[24], flags = -1, nw = -1, step;
        memset(&u, 0, sizeof u);
        u.sound = sound; u.t0 = t0; u.rir = rir; u.dis_sound = dis_sound; u.dis_rir = dis_rir;
        for (step = 0; step < 2; ++step) {
            int i;
            CHECK(ss_ctx_plan(ctx, &u, 3, desc, &flags, &nw, NULL, 0));
            printf("step %d flags %d new_windows %d\n", step, flags, nw);
            for (i = 0; i < 3; ++i)
                printf("unit %d: %d %d %d %d | %d %d %d %d\n", i, desc[8 * i], desc[8 * i + 1], desc[8 * i + 2],
                       desc[8 * i + 3], desc[8 * i + 4], desc[8 * i + 5], desc[8 * i + 6], desc[8 * i + 7]);
        }
    }
    {
        long long st[8];
        CHECK(ss_ctx_stats(ctx, st));
        printf("hits %lld misses %lld resident %lld\n", st[0], st[1], st[5]);
    }
    CHECK(ss_ctx_destroy(ctx));
    return 0;
}

/* A vector env's state columns -> unit columns, as a C host would hand them over each step (host only). */
static int sims_only(void) {
    ss_ctx* ctx = NULL;
    /* 3 envs in one scene of 3 nodes; pair (receiver 1, source 2) sits in bank rows 40..43, (2, 2) in 8..11, rest absent */
    long long sound[3] = {0, 1, 1}, audio_index[3] = {0, 1, 4}, step_count[3] = {3, 3, 9}, duration[3] = {5, 5, 5};
    long long recv[3] = {1, 2, 0}, src[3] = {2, 2, 1}, rot[3] = {0, 90, 0}, scene[3] = {0, 0, 0};
    int index_flat[9] = {-1, -1, -1, -1, -1, 40, -1, -1, 8};
    long long off[1] = {0}, dim[1] = {3};
    int units[15], miss[3], n_miss = -1, i, step;
    ss_sim_columns c;
    CHECK(ss_ctx_create(&ctx, 16000, 16000, SS_PAD_REFLECT, 0, 8));
    if (ss_ctx_add_source_len(ctx, 16000) != 0 || ss_ctx_add_source_len(ctx, 80000) != 1) return 11;
    memset(&c, 0, sizeof c);
    c.sound = sound; c.audio_index = audio_index; c.step_count = step_count; c.duration = duration;
    c.recv = recv; c.src = src; c.rot = rot; c.scene = scene;
    c.index_flat = index_flat; c.index_off = off; c.index_dim = dim; c.n_scenes = 1; c.azimuths = 4;
    for (step = 0; step < 2; ++step) {
        CHECK(ss_ctx_sims_units(ctx, &c, 3, units, miss, &n_miss));
        printf("step %d misses %d\n", step, n_miss);
        for (i = 0; i < 3; ++i)
            printf("env %d: sound %d t0 %d rir %d audio_index %lld\n", i, units[i], units[3 + i], units[6 + i], audio_index[i]);
        recv[2] = 0; src[2] = 0; step_count[2] = 1;      /* env 2 starts a new episode at a pair that is not resident */
    }
    CHECK(ss_ctx_destroy(ctx));
    return 0;
}

#ifdef WITH_HIP
static int rd(void* p, size_t n, FILE* f) { return fread(p, 1, n, f) == n ? 0 : 1; }

static int observe(const char* in_path, const char* out_path) {
    FILE* f = fopen(in_path, "rb");
    int sr, n_src, R, cap, n, i, t4;
    int *src_len, *rir_len, *sound, *t0, *rir_idx;
    float *rir, *d_rir = NULL, *d_ag = NULL, *d_sg = NULL, *ag, *sg;
    int* d_len = NULL;
    size_t ag_n, sg_n;
    ss_ctx* ctx = NULL;
    ss_units u;
    if (!f) return 20;
    if (rd(&sr, 4, f) || rd(&n_src, 4, f)) return 21;
    CHECK(ss_ctx_create(&ctx, sr, sr, SS_PAD_REFLECT, 0, 0));
    src_len = (int*)malloc(4 * (size_t)n_src);
    if (rd(src_len, 4 * (size_t)n_src, f)) return 22;
    for (i = 0; i < n_src; ++i) {
        float* clip = (float*)malloc(4 * (size_t)src_len[i]);
        if (rd(clip, 4 * (size_t)src_len[i], f)) return 23;
        if (ss_ctx_add_source(ctx, clip, src_len[i], 0) != i) return 24;      /* host clip -> the library's device bank */
        free(clip);
    }
    if (rd(&R, 4, f) || rd(&cap, 4, f)) return 25;
    rir_len = (int*)malloc(4 * (size_t)R);
    rir = (float*)malloc(4 * (size_t)R * 2 * cap);
    if (rd(rir_len, 4 * (size_t)R, f) || rd(rir, 4 * (size_t)R * 2 * cap, f)) return 26;
    if (rd(&n, 4, f)) return 27;
    sound = (int*)malloc(4 * (size_t)n); t0 = (int*)malloc(4 * (size_t)n); rir_idx = (int*)malloc(4 * (size_t)n);
    if (rd(sound, 4 * (size_t)n, f) || rd(t0, 4 * (size_t)n, f) || rd(rir_idx, 4 * (size_t)n, f)) return 28;
    fclose(f);
    t4 = ((1 + sr / 160) + 3) / 4;
    ag_n = (size_t)n * 2 * sr; sg_n = (size_t)n * 65 * t4 * 2;
    if (hipMalloc((void**)&d_rir, 4 * (size_t)R * 2 * cap) != hipSuccess || hipMalloc((void**)&d_len, 4 * (size_t)R) != hipSuccess ||
        hipMalloc((void**)&d_ag, 4 * ag_n) != hipSuccess || hipMalloc((void**)&d_sg, 4 * sg_n) != hipSuccess) return 29;
    if (hipMemcpy(d_rir, rir, 4 * (size_t)R * 2 * cap, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_len, rir_len, 4 * (size_t)R, hipMemcpyHostToDevice) != hipSuccess) return 30;
    CHECK(ss_ctx_set_rir_bank(ctx, d_rir, d_len, 2LL * cap, cap, 1, cap));
    memset(&u, 0, sizeof u);
    u.sound = sound; u.t0 = t0; u.rir = rir_idx;
    CHECK(ss_ctx_observe(ctx, &u, n, d_ag, d_sg, NULL));                       /* one call = one vector step */
    ag = (float*)malloc(4 * ag_n); sg = (float*)malloc(4 * sg_n);
    if (hipMemcpy(ag, d_ag, 4 * ag_n, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(sg, d_sg, 4 * sg_n, hipMemcpyDeviceToHost) != hipSuccess) return 31;
    f = fopen(out_path, "wb");
    if (!f || fwrite(ag, 4, ag_n, f) != ag_n || fwrite(sg, 4, sg_n, f) != sg_n) return 32;
    fclose(f);
    CHECK(ss_ctx_destroy(ctx));
    return 0;
}
#endif

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "plan") == 0) return plan_only();
    if (argc >= 2 && strcmp(argv[1], "sims") == 0) return sims_only();
#ifdef WITH_HIP
    if (argc >= 4 && strcmp(argv[1], "observe") == 0) return observe(argv[2], argv[3]);
#endif
    fprintf(stderr, "usage: ctx_client plan | sims | observe in.bin out.bin\n");
    return 2;
}
