/* A plain-C consumer of include/boxtree_hip.h: no Python, no torch.  Builds a tree
 * over pseudo-random points held in device memory and prints what the Python layer's
 * TreeBuilder returns for the same points (tests/test_gpu_cabi.py compares the two).
 *
 *   cabi_tree <dims> <n> <max_particles_in_box> <seed>
 *
 * The root box is computed the way the host code of tree_build.py:456-476 does
 * (boxtree_amd/tree_build.py mirrors it): extent = max axis range * (1 + 1e-4),
 * bbox_max = bbox_min + extent, all in double. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "boxtree_hip.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_BT(e) do { int s_ = (e); if (s_ != 0) { \
    fprintf(stderr, "%s:%d: boxtree error %d: %s\n", __FILE__, __LINE__, s_, \
            bt_last_error_string()); return 3; } } while (0)

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s dims n mpb seed\n", argv[0]); return 1; }
    const int dims = atoi(argv[1]);
    const int64_t n = atoll(argv[2]);
    const int mpb = atoi(argv[3]);
    uint64_t seed = (uint64_t) atoll(argv[4]);
    if (dims < 1 || dims > 3 || n < 1) return 1;

    double *host[3] = {0, 0, 0};
    void *dev[3] = {0, 0, 0};
    for (int ax = 0; ax < dims; ++ax) {
        host[ax] = (double *) malloc((size_t) n * sizeof(double));
        for (int64_t i = 0; i < n; ++i)                       /* uniform in [0, 1) */
            host[ax][i] = (double) (splitmix64(&seed) >> 11) * (1.0 / 9007199254740992.0);
        CHECK_HIP(hipMalloc(&dev[ax], (size_t) n * sizeof(double)));
        CHECK_HIP(hipMemcpy(dev[ax], host[ax], (size_t) n * sizeof(double), hipMemcpyHostToDevice));
    }

    if (bt_abi_version() != BT_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 4; }
    bt_context *ctx = NULL;
    CHECK_BT(bt_create(0, NULL, &ctx));

    double bmin[3], bmax[3];
    CHECK_BT(bt_bbox(ctx, dims, BT_F64, (const void *const *) dev, NULL, n, bmin, bmax));
    double extent = 0;
    for (int ax = 0; ax < dims; ++ax)
        if (bmax[ax] - bmin[ax] > extent) extent = bmax[ax] - bmin[ax];
    extent *= 1 + 1e-4;

    bt_tree_params p;
    memset(&p, 0, sizeof(p));
    p.dims = dims;
    p.coord_kind = BT_F64;
    p.nsources = n;
    p.ntargets = -1;                       /* sources are the targets */
    for (int ax = 0; ax < dims; ++ax) {
        p.sources[ax] = dev[ax];
        p.bbox_min[ax] = bmin[ax];
        p.bbox_max[ax] = bmin[ax] + extent;
    }
    p.max_leaf_refine_weight = mpb;
    p.kind = BT_KIND_ADAPTIVE;
    p.extent_norm = BT_NORM_NONE;
    p.root_extent = extent;

    bt_tree_sizes sz;
    CHECK_BT(bt_tree_build(ctx, &p, &sz));

    const int C = 1 << dims;
    bt_tree_arrays o;
    memset(&o, 0, sizeof(o));
    int32_t *i32[8];
    size_t nb = (size_t) sz.nboxes, al = (size_t) sz.aligned_nboxes;
    CHECK_HIP(hipMalloc((void **) &o.user_source_ids, (size_t) n * 4));
    CHECK_HIP(hipMalloc((void **) &o.sorted_target_ids, (size_t) n * 4));
    for (int ax = 0; ax < dims; ++ax) CHECK_HIP(hipMalloc(&o.sources[ax], (size_t) n * 8));
    CHECK_HIP(hipMalloc((void **) &o.box_source_starts, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_source_counts_nonchild, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_source_counts_cumul, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_parent_ids, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_child_ids, (size_t) C * al * 4));
    CHECK_HIP(hipMalloc(&o.box_centers, (size_t) dims * al * 8));
    CHECK_HIP(hipMalloc((void **) &o.box_levels, nb));
    CHECK_HIP(hipMalloc((void **) &o.box_flags, nb));
    CHECK_HIP(hipMalloc(&o.box_source_bounding_box_min, (size_t) dims * al * 8));
    CHECK_HIP(hipMalloc(&o.box_source_bounding_box_max, (size_t) dims * al * 8));
    (void) i32;
    CHECK_BT(bt_tree_export(ctx, &o));

    /* a digest the Python side can recompute: sums over a few arrays */
    int32_t *h_cumul = (int32_t *) malloc(nb * 4), *h_ids = (int32_t *) malloc((size_t) n * 4);
    uint8_t *h_levels = (uint8_t *) malloc(nb);
    CHECK_HIP(hipMemcpy(h_cumul, o.box_source_counts_cumul, nb * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_ids, o.user_source_ids, (size_t) n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_levels, o.box_levels, nb, hipMemcpyDeviceToHost));
    uint64_t d_cumul = 0, d_ids = 0, d_levels = 0;
    for (size_t b = 0; b < nb; ++b) { d_cumul += (uint64_t) h_cumul[b] * (b + 1); d_levels += h_levels[b]; }
    for (int64_t i = 0; i < n; ++i) d_ids += (uint64_t) h_ids[i] * (uint64_t) (i % 1000003 + 1);
    printf("nboxes %lld nlevels %d aligned %lld root_extent %.17g cumul %llu ids %llu levels %llu\n",
           (long long) sz.nboxes, sz.nlevels, (long long) sz.aligned_nboxes, extent,
           (unsigned long long) d_cumul, (unsigned long long) d_ids,
           (unsigned long long) d_levels);
    bt_destroy(ctx);
    return 0;
}
