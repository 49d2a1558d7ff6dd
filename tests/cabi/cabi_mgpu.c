/* A plain-C consumer of the multi-GPU entries of include/boxtree_hip.h: no Python, no
 * torch.  NRANKS ranks run as threads of this process (bt_mgpu_local_group_create: a test
 * box has one GPU, which RCCL will not share between ranks; with an RCCL communicator per
 * process the calls are the same), every rank with a context of its own:
 *
 *   bt_mgpu_exchange -> bt_tree_build / bt_tree_export -> bt_mgpu_number
 *   -> bt_mgpu_let_build / bt_mgpu_let_export
 *
 *   cabi_mgpu <nranks> <dims> <n_per_rank> <max_particles_in_box> <seed> [n_targets_per_rank]
 *
 * Rank r draws its chunk from the stream seeded with seed + r (splitmix64, uniform).  With
 * n_targets_per_rank > 0 every rank also has separate targets with extents -- coordinates from
 * the stream seeded with seed + 1000 + r, radii 2^-4 * 2^(-12 u) from the same stream after them,
 * stick-out factor 0.25, l^inf --: the sharded build of particles with extents from plain C.  One
 * After the build every rank asks the library who its particles are (bt_mgpu_global_ids), checks
 * on the host that received particle j carries the x coordinate of global particle ids[j] (the
 * draw of the stream of the rank that id belongs to) and that an array routed to the owners and
 * back (bt_mgpu_route) is unchanged, and reports a digest of the global user_source_ids of its
 * slice of the tree order.  One
 * line per rank: what it owns and where its boxes sit in the global tree;
 * tests/test_gpu_cabi.py compares the global figures with the tree the Python layer
 * builds on one GPU from all chunks. */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
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

typedef struct {
    int rank, nranks, dims, mpb;
    int64_t n, nt;
    uint64_t seed;
    void *group;
    int status;
    /* results */
    int64_t n_owned, nboxes_local, nboxes_global, let_nboxes, halo_in, source_offset, nsources_global;
    int64_t nt_owned, target_offset, ntargets_global;
    int32_t nlevels_global;
    uint64_t digest_ids;           /* sum of the global numbers of the rank's deep boxes */
    uint64_t digest_user_ids;      /* sum over the rank's tree positions p of (source_offset + p + 1) * */
                                   /* (global user id of the source at p), mod 2^64                    */
    int64_t chunk_offset;
} rank_args;

static int run_rank(rank_args *a)
{
    const int dims = a->dims;
    const int64_t n = a->n;
    CHECK_HIP(hipSetDevice(0));
    uint64_t seed = a->seed + (uint64_t) a->rank;
    void *dev[3] = {0, 0, 0};
    for (int ax = 0; ax < dims; ++ax) {
        double *host = (double *) malloc((size_t) n * sizeof(double));
        for (int64_t i = 0; i < n; ++i)
            host[i] = (double) (splitmix64(&seed) >> 11) * (1.0 / 9007199254740992.0);
        CHECK_HIP(hipMalloc(&dev[ax], (size_t) n * sizeof(double)));
        CHECK_HIP(hipMemcpy(dev[ax], host, (size_t) n * sizeof(double), hipMemcpyHostToDevice));
        free(host);
    }
    const int64_t nt = a->nt;
    void *tdev[3] = {0, 0, 0}, *rdev = NULL;
    if (nt > 0) {
        uint64_t tseed = a->seed + 1000u + (uint64_t) a->rank;
        double *host = (double *) malloc((size_t) nt * sizeof(double));
        for (int ax = 0; ax < dims; ++ax) {
            for (int64_t i = 0; i < nt; ++i)
                host[i] = (double) (splitmix64(&tseed) >> 11) * (1.0 / 9007199254740992.0);
            CHECK_HIP(hipMalloc(&tdev[ax], (size_t) nt * sizeof(double)));
            CHECK_HIP(hipMemcpy(tdev[ax], host, (size_t) nt * sizeof(double), hipMemcpyHostToDevice));
        }
        for (int64_t i = 0; i < nt; ++i) {
            const double u = (double) (splitmix64(&tseed) >> 11) * (1.0 / 9007199254740992.0);
            /* 2^-4 * 2^(-12 u) without libm: 2^-(4 + k) * (1 - f/2), k = floor(12 u), f its fraction
             * (a fixed, reproducible function of u; the Python side evaluates the same expression) */
            const int k = (int) (12.0 * u);
            const double f = 12.0 * u - (double) k;
            host[i] = (1.0 / (double) (1u << (4 + k))) * (1.0 - 0.5 * f);
        }
        CHECK_HIP(hipMalloc(&rdev, (size_t) nt * sizeof(double)));
        CHECK_HIP(hipMemcpy(rdev, host, (size_t) nt * sizeof(double), hipMemcpyHostToDevice));
        free(host);
    }
    bt_context *ctx = NULL;
    CHECK_BT(bt_create(0, NULL, &ctx));
    bt_mgpu_comm *comm = NULL;
    CHECK_BT(bt_mgpu_comm_local(a->group, a->rank, &comm));

    /* steps 1-3 */
    bt_mgpu_params mp;
    memset(&mp, 0, sizeof(mp));
    mp.dims = dims; mp.coord_kind = BT_F64; mp.n = n; mp.max_particles_in_box = a->mpb;
    for (int ax = 0; ax < dims; ++ax) mp.coords[ax] = dev[ax];
    if (nt > 0) {
        mp.ntargets = nt;
        for (int ax = 0; ax < dims; ++ax) mp.targets[ax] = tdev[ax];
        mp.target_radii = rdev; mp.stick_out_factor = 0.25; mp.extent_norm = BT_NORM_LINF;
    }
    bt_mgpu_shard sh;
    CHECK_BT(bt_mgpu_exchange(ctx, comm, &mp, &sh));

    /* step 4 */
    bt_tree_params tp;
    memset(&tp, 0, sizeof(tp));
    tp.dims = dims; tp.coord_kind = BT_F64; tp.nsources = sh.n_owned; tp.ntargets = -1;
    tp.max_leaf_refine_weight = a->mpb; tp.kind = BT_KIND_ADAPTIVE; tp.extent_norm = BT_NORM_NONE;
    tp.root_extent = sh.root_extent; tp.top_level = sh.top_level; tp.top_cell_prefix = sh.top_cell_prefix;
    tp.source_stride = dims;
    for (int ax = 0; ax < dims; ++ax) {
        tp.sources[ax] = (const double *) sh.points + ax;
        tp.bbox_min[ax] = sh.bbox_min[ax]; tp.bbox_max[ax] = sh.bbox_max[ax];
    }
    if (nt > 0) {
        /* the received targets: records of dims + 1 values, radii once more as a dense array;
         * the top of the global tree as arrivals / stayers per top box */
        tp.ntargets = sh.n_owned_targets;
        tp.target_stride = sh.target_record_len;
        for (int ax = 0; ax < dims; ++ax) tp.targets[ax] = (const double *) sh.target_points + ax;
        tp.target_radii = sh.target_radii;
        tp.stick_out_factor = 0.25; tp.extent_norm = BT_NORM_LINF;
        tp.top_box_arrive = sh.top_box_arrive; tp.top_box_stay = sh.top_box_stay;
    }
    bt_tree_sizes sz;
    CHECK_BT(bt_tree_build(ctx, &tp, &sz));
    const int C = 1 << dims;
    const size_t nb = (size_t) sz.nboxes, al = (size_t) sz.aligned_nboxes, no = (size_t) (sh.n_owned > 0 ? sh.n_owned : 1);
    bt_tree_arrays o;
    memset(&o, 0, sizeof(o));
    CHECK_HIP(hipMalloc((void **) &o.user_source_ids, no * 4));
    CHECK_HIP(hipMalloc((void **) &o.sorted_target_ids, no * 4));
    for (int ax = 0; ax < dims; ++ax) CHECK_HIP(hipMalloc(&o.sources[ax], no * 8));
    CHECK_HIP(hipMalloc((void **) &o.box_source_starts, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_source_counts_nonchild, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_source_counts_cumul, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_parent_ids, nb * 4));
    CHECK_HIP(hipMalloc((void **) &o.box_child_ids, (size_t) C * al * 4));
    CHECK_HIP(hipMalloc(&o.box_centers, (size_t) dims * al * 8));
    CHECK_HIP(hipMalloc((void **) &o.box_levels, nb));
    CHECK_HIP(hipMalloc((void **) &o.box_flags, nb));
    CHECK_HIP(hipMalloc((void **) &o.box_subtree_sizes, nb * 4));
    CHECK_HIP(hipMalloc(&o.box_source_bounding_box_min, (size_t) dims * al * 8));
    CHECK_HIP(hipMalloc(&o.box_source_bounding_box_max, (size_t) dims * al * 8));
    if (nt > 0) {
        const size_t nto = (size_t) (sh.n_owned_targets > 0 ? sh.n_owned_targets : 1);
        for (int ax = 0; ax < dims; ++ax) CHECK_HIP(hipMalloc(&o.targets[ax], nto * 8));
        CHECK_HIP(hipMalloc(&o.target_radii, nto * 8));
        CHECK_HIP(hipMalloc((void **) &o.box_target_starts, nb * 4));
        CHECK_HIP(hipMalloc((void **) &o.box_target_counts_nonchild, nb * 4));
        CHECK_HIP(hipMalloc((void **) &o.box_target_counts_cumul, nb * 4));
        CHECK_HIP(hipMalloc(&o.box_target_bounding_box_min, (size_t) dims * al * 8));
        CHECK_HIP(hipMalloc(&o.box_target_bounding_box_max, (size_t) dims * al * 8));
    }
    CHECK_BT(bt_tree_export(ctx, &o));
    CHECK_BT(bt_synchronize(ctx));

    /* particle identity: who are the particles this rank received? */
    uint64_t id_digest = 0;
    {
        int32_t *d_ids = NULL, *h_ids = (int32_t *) malloc(no * 4), *h_usid = (int32_t *) malloc(no * 4);
        double *d_x = NULL, *d_back = NULL, *h_x = (double *) malloc(no * 8), *h_rec = (double *) malloc(no * 8 * (size_t) dims);
        double *h_back = (double *) malloc((size_t) n * 8), *h_mine = (double *) malloc((size_t) n * 8);
        CHECK_HIP(hipMalloc((void **) &d_ids, no * 4));
        CHECK_HIP(hipMalloc((void **) &d_x, no * 8));
        CHECK_HIP(hipMalloc((void **) &d_back, (size_t) n * 8));
        CHECK_BT(bt_mgpu_global_ids(ctx, comm, 0, 4, d_ids));
        CHECK_BT(bt_mgpu_route(ctx, comm, 0, BT_ROUTE_TO_OWNERS, 8, dev[0], d_x));
        CHECK_BT(bt_mgpu_route(ctx, comm, 0, BT_ROUTE_TO_CALLERS, 8, d_x, d_back));
        CHECK_BT(bt_synchronize(ctx));
        CHECK_HIP(hipMemcpy(h_ids, d_ids, (size_t) sh.n_owned * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_x, d_x, (size_t) sh.n_owned * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_rec, sh.points, (size_t) sh.n_owned * 8 * (size_t) dims, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_usid, o.user_source_ids, (size_t) sh.n_owned * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_back, d_back, (size_t) n * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_mine, dev[0], (size_t) n * 8, hipMemcpyDeviceToHost));
        if (memcmp(h_back, h_mine, (size_t) n * 8) != 0) { fprintf(stderr, "rank %d: route round trip differs\n", a->rank); return 92; }
        if (sh.source_chunk_offset != (int64_t) a->rank * n || sh.n_global_sources != (int64_t) a->nranks * n) {
            fprintf(stderr, "rank %d: chunk offset %lld of %lld\n", a->rank, (long long) sh.source_chunk_offset,
                    (long long) sh.n_global_sources);
            return 93;
        }
        /* x of every chunk's particles, as their ranks drew them */
        double *all_x = (double *) malloc((size_t) a->nranks * (size_t) n * 8);
        for (int q = 0; q < a->nranks; ++q) {
            uint64_t sq = a->seed + (uint64_t) q;
            for (int64_t i = 0; i < n; ++i) all_x[(size_t) q * (size_t) n + (size_t) i] = (double) (splitmix64(&sq) >> 11) * (1.0 / 9007199254740992.0);
        }
        for (int64_t j = 0; j < sh.n_owned; ++j) {
            const int32_t g = h_ids[j];
            if (g < 0 || (int64_t) g >= (int64_t) a->nranks * n || all_x[g] != h_x[j] || h_rec[(size_t) j * (size_t) dims] != h_x[j]) {
                fprintf(stderr, "rank %d: received particle %lld is not global particle %d\n", a->rank, (long long) j, g);
                return 94;
            }
        }
        /* (filled in with the global source offset after bt_mgpu_number) */
        for (int64_t pidx = 0; pidx < sh.n_owned; ++pidx)
            id_digest += (uint64_t) (pidx + 1) * (uint64_t) h_ids[h_usid[pidx]];
        a->digest_user_ids = 0;
        for (int64_t pidx = 0; pidx < sh.n_owned; ++pidx) a->digest_user_ids += (uint64_t) h_ids[h_usid[pidx]];
        free(all_x); free(h_ids); free(h_usid); free(h_x); free(h_rec); free(h_back); free(h_mine);
        (void) hipFree(d_ids); (void) hipFree(d_x); (void) hipFree(d_back);
    }

    /* step 5 */
    bt_mgpu_local_tree lt;
    memset(&lt, 0, sizeof(lt));
    lt.dims = dims; lt.coord_kind = BT_F64; lt.nboxes = sz.nboxes; lt.aligned_nboxes = sz.aligned_nboxes;
    lt.nlevels = sz.nlevels; lt.level_start_box_nrs = sz.level_start_box_nrs;
    lt.box_centers = o.box_centers; lt.box_levels = o.box_levels; lt.box_flags = o.box_flags;
    lt.nsources = sh.n_owned; lt.ntargets = nt > 0 ? sh.n_owned_targets : sh.n_owned;
    lt.box_subtree_sizes = o.box_subtree_sizes;     /* every rank: the LET then comes with sizes */
    if (nt > 0) {
        lt.box_target_bounding_box_min = o.box_target_bounding_box_min;
        lt.box_target_bounding_box_max = o.box_target_bounding_box_max;
        lt.box_source_counts_cumul = o.box_source_counts_cumul;
    }
    int32_t *d_box_ids = NULL;
    CHECK_HIP(hipMalloc((void **) &d_box_ids, nb * 4));
    bt_mgpu_numbering num;
    CHECK_BT(bt_mgpu_number(ctx, comm, &lt, d_box_ids, &num));

    /* step 6 */
    bt_mgpu_let_sizes ls;
    CHECK_BT(bt_mgpu_let_build(ctx, comm, &lt, d_box_ids, &num, 1, &ls));
    bt_mgpu_let_arrays la;
    memset(&la, 0, sizeof(la));
    const size_t lb = (size_t) ls.nboxes, lal = (size_t) ls.aligned_nboxes;
    CHECK_HIP(hipMalloc(&la.box_centers, (size_t) dims * lal * 8));
    CHECK_HIP(hipMalloc((void **) &la.box_parent_ids, lb * 4));
    CHECK_HIP(hipMalloc((void **) &la.box_child_ids, (size_t) C * lal * 4));
    CHECK_HIP(hipMalloc((void **) &la.box_levels, lb));
    CHECK_HIP(hipMalloc((void **) &la.box_flags, lb));
    CHECK_HIP(hipMalloc((void **) &la.global_box_ids, lb * 4));
    CHECK_HIP(hipMalloc((void **) &la.target_boxes_mask, lb));
    if (nt > 0) {
        CHECK_HIP(hipMalloc(&la.box_target_bounding_box_min, (size_t) dims * lal * 8));
        CHECK_HIP(hipMalloc(&la.box_target_bounding_box_max, (size_t) dims * lal * 8));
        CHECK_HIP(hipMalloc((void **) &la.box_source_counts_cumul, lb * 4));
    }
    if (!ls.has_subtree_sizes) { fprintf(stderr, "rank %d: the LET has no subtree sizes\n", a->rank); return 90; }
    CHECK_HIP(hipMalloc((void **) &la.box_subtree_sizes, lb * 4));
    CHECK_BT(bt_mgpu_let_export(ctx, &la));
    CHECK_BT(bt_synchronize(ctx));
    {
        int32_t root_size = 0;      /* the root's subtree is the whole LET */
        CHECK_HIP(hipMemcpy(&root_size, la.box_subtree_sizes, 4, hipMemcpyDeviceToHost));
        if ((int64_t) root_size != ls.nboxes) {
            fprintf(stderr, "rank %d: root subtree size %d, LET boxes %lld\n", a->rank, root_size, (long long) ls.nboxes);
            return 91;
        }
    }

    /* digest: global numbers of the boxes this rank owns below the shared top levels */
    int32_t *h_ids = (int32_t *) malloc(nb * 4);
    uint8_t *h_lev = (uint8_t *) malloc(nb);
    CHECK_HIP(hipMemcpy(h_ids, d_box_ids, nb * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_lev, o.box_levels, nb, hipMemcpyDeviceToHost));
    uint64_t dg = 0;
    for (size_t b = 0; b < nb; ++b)
        if (h_lev[b] > sh.top_level) dg += (uint64_t) h_ids[b];
    a->n_owned = sh.n_owned; a->nboxes_local = sz.nboxes; a->nboxes_global = num.nboxes;
    a->nlevels_global = num.nlevels; a->let_nboxes = ls.nboxes; a->halo_in = ls.halo_boxes_received;
    a->source_offset = num.source_offset; a->nsources_global = num.nsources; a->digest_ids = dg;
    a->nt_owned = sh.n_owned_targets; a->target_offset = num.target_offset; a->ntargets_global = num.ntargets;
    /* sum_p (source_offset + p + 1) * id(p) = id_digest + source_offset * sum_p id(p) */
    a->digest_user_ids = id_digest + (uint64_t) num.source_offset * a->digest_user_ids;
    a->chunk_offset = sh.source_chunk_offset;
    bt_mgpu_comm_destroy(comm);
    bt_destroy(ctx);
    return 0;
}

static void *thread_main(void *p)
{
    rank_args *a = (rank_args *) p;
    a->status = run_rank(a);
    if (a->status != 0) {           /* the other ranks would wait for this one forever */
        fprintf(stderr, "rank %d failed with status %d\n", a->rank, a->status);
        exit(a->status);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc != 6 && argc != 7) { fprintf(stderr, "usage: %s nranks dims n_per_rank mpb seed [n_targets_per_rank]\n", argv[0]); return 1; }
    const int64_t nt = argc == 7 ? atoll(argv[6]) : 0;
    const int nranks = atoi(argv[1]), dims = atoi(argv[2]), mpb = atoi(argv[4]);
    const int64_t n = atoll(argv[3]);
    const uint64_t seed = (uint64_t) atoll(argv[5]);
    if (nranks < 1 || nranks > 64 || dims < 1 || dims > 3 || n < 1) return 1;
    if (bt_abi_version() != BT_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 4; }
    void *group = NULL;
    CHECK_BT(bt_mgpu_local_group_create(nranks, &group));
    rank_args *args = (rank_args *) calloc((size_t) nranks, sizeof(rank_args));
    pthread_t *th = (pthread_t *) calloc((size_t) nranks, sizeof(pthread_t));
    for (int r = 0; r < nranks; ++r) {
        args[r].rank = r; args[r].nranks = nranks; args[r].dims = dims; args[r].mpb = mpb;
        args[r].n = n; args[r].nt = nt; args[r].seed = seed; args[r].group = group;
        pthread_create(&th[r], NULL, thread_main, &args[r]);
    }
    for (int r = 0; r < nranks; ++r) pthread_join(th[r], NULL);
    for (int r = 0; r < nranks; ++r)
        printf("rank %d owned %lld source_offset %lld nboxes_local %lld nboxes_global %lld "
               "nlevels_global %d nsources_global %lld let_nboxes %lld halo_in %lld deep_ids %llu "
               "targets_owned %lld target_offset %lld ntargets_global %lld user_id_digest %llu chunk_offset %lld\n",
               r, (long long) args[r].n_owned, (long long) args[r].source_offset,
               (long long) args[r].nboxes_local, (long long) args[r].nboxes_global,
               args[r].nlevels_global, (long long) args[r].nsources_global,
               (long long) args[r].let_nboxes, (long long) args[r].halo_in,
               (unsigned long long) args[r].digest_ids, (long long) args[r].nt_owned,
               (long long) args[r].target_offset, (long long) args[r].ntargets_global,
               (unsigned long long) args[r].digest_user_ids, (long long) args[r].chunk_offset);
    bt_mgpu_local_group_destroy(group);
    return 0;
}
