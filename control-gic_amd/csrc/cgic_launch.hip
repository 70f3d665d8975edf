// cgic_launch.hip -- host-side launcher for several captured hipGraphs at once (pipeline.LaneStream: one graph per lane).
//
// A lane's work is one hipGraphLaunch on its own stream; from Python the four launches of a submit are four interpreter round
// trips (stream context manager + binding + hipGraphLaunch, ~25-30 us each), so lane 3 starts ~100 us after lane 0 -- a
// visible share of a 20-step window of ~800 us.  Here the launches are one C call, back to back on the calling thread.
// (Round 3 also had one persistent worker thread per lane: the runtime serialises the launches, the window got no shorter,
// and the pool was not reentrant -- removed in round 4, NOTES.md.)
#include "cgic_common.h"

extern "C" int cgic_launch_graphs(void *const *graph_execs, void *const *streams, int n)
{
    CGIC_REQUIRE(n >= 0 && (n == 0 || (graph_execs && streams)), CGIC_ERR_INVALID, "launch_graphs: NULL argument");
    for (int i = 0; i < n; ++i) CGIC_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_execs[i], (hipStream_t)streams[i]));
    return CGIC_OK;
}

// ---- launch groups ---------------------------------------------------------------------------------------------------
// (cgic_common.h has the idea.)  State is per thread: the calls of a group come from the thread that opened it.
#include <map>
#include <memory>

namespace cgic {

struct GroupState {
    int ngroups = 0, cur = 0;
    double share[kMaxGroups] = {1.0, 1.0, 1.0, 1.0};
    std::vector<GroupRec> rec[kMaxGroups];
};
static thread_local std::unique_ptr<GroupState> t_group;

static GroupedLauncher *grouped_table()
{
    static GroupedLauncher table[KID_COUNT] = {};
    return table;
}
GroupedRegistrar::GroupedRegistrar(int kid, GroupedLauncher fn) { grouped_table()[kid] = fn; }

bool group_recording() { return (bool)t_group; }
double group_cu_share() { return t_group ? t_group->share[t_group->cur] : 1.0; }

int group_record(int kid, dim3 grid, dim3 block, size_t lds, const void *args, size_t bytes, hipStream_t stream, std::function<int()> launch)
{
    GroupState *g = t_group.get();
    CGIC_REQUIRE(g, CGIC_ERR_INVALID, "group_record outside a group");
    GroupRec r;
    r.kid = kid; r.stream = stream; r.grid = grid; r.block = block; r.lds = lds;
    r.args.assign((const unsigned char *)args, (const unsigned char *)args + bytes);
    r.launch = std::move(launch);
    g->rec[g->cur].push_back(std::move(r));
    return CGIC_OK;
}

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_group_max(void) { return kMaxGroups; }

extern "C" int cgic_group_begin(int ngroups, const double *shares)
{
    CGIC_REQUIRE(!t_group, CGIC_ERR_INVALID, "group_begin: a group is already open on this thread");
    CGIC_REQUIRE(ngroups >= 1 && ngroups <= kMaxGroups, CGIC_ERR_INVALID, "group_begin: %d groups (1..%d)", ngroups, kMaxGroups);
    std::unique_ptr<GroupState> g(new GroupState);
    g->ngroups = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        g->share[i] = shares ? shares[i] : 1.0 / ngroups;
        CGIC_REQUIRE(g->share[i] > 0.0 && g->share[i] <= 1.0, CGIC_ERR_INVALID, "group_begin: share[%d] = %g", i, g->share[i]);
    }
    t_group = std::move(g);
    return CGIC_OK;
}

extern "C" int cgic_group_select(int group)
{
    CGIC_REQUIRE(t_group, CGIC_ERR_INVALID, "group_select: no group is open on this thread");
    CGIC_REQUIRE(group >= 0 && group < t_group->ngroups, CGIC_ERR_INVALID, "group_select: group %d of %d", group, t_group->ngroups);
    t_group->cur = group;
    return CGIC_OK;
}

extern "C" void cgic_group_abort(void) { t_group.reset(); }

extern "C" int cgic_group_launch(cgic_stream_t stream)
{
    CGIC_REQUIRE(t_group, CGIC_ERR_INVALID, "group_launch: no group is open on this thread");
    std::unique_ptr<GroupState> g = std::move(t_group);          // closed whatever happens below
    size_t depth = 0;
    for (int i = 0; i < g->ngroups; ++i) depth = g->rec[i].size() > depth ? g->rec[i].size() : depth;
    // every record was made for one stream -- its ticket slots were taken from that stream's ring, its single-launch fallback
    // goes to it -- and the grouped launches go to `stream`: they have to be the same one, or a dependent chain would be split over
    // two streams (checked before anything is enqueued)
    for (int i = 0; i < g->ngroups; ++i)
        for (const GroupRec &r : g->rec[i])
            CGIC_REQUIRE(r.stream == (hipStream_t)stream, CGIC_ERR_INVALID,
                         "group_launch: a launch of group %d was recorded for another stream than the one given here", i);
    int launches = 0;
    for (size_t j = 0; j < depth; ++j) {
        const GroupRec *recs[kMaxGroups];
        int n = 0;
        for (int i = 0; i < g->ngroups; ++i)
            if (j < g->rec[i].size()) recs[n++] = &g->rec[i][j];
        bool same = n >= 2 && recs[0]->kid != KID_NONE && grouped_table()[recs[0]->kid] != nullptr;
        for (int i = 1; same && i < n; ++i)
            same = recs[i]->kid == recs[0]->kid && recs[i]->block.x == recs[0]->block.x && recs[i]->block.y == 1 && recs[i]->block.z == 1;
        if (same) {
            int rc = grouped_table()[recs[0]->kid](recs, n, (hipStream_t)stream);
            if (rc) return rc;
            ++launches;
            continue;
        }
        for (int i = 0; i < n; ++i) {
            int rc = recs[i]->launch();
            if (rc) return rc;
            ++launches;
        }
    }
    return launches;          // >= 0: launches issued
}

// One call per call the reference makes (include/cgic_hip.h, section H'): the four entry points of the hot path, in order, on one stream.
extern "C" int cgic_compress_image(const cgic_table *t, const float *codebook, int K, int e_dim, const void *prepared, int64_t B, int64_t H,
                                   int64_t W, double coarse_ratio, double medium_ratio, float beta, int legacy, const float *bins, int nbins,
                                   float sigma, int decoder, const cgic_image_io *io, int *mode_out, cgic_stream_t stream)
{
    CGIC_REQUIRE(io && io->x && io->z && io->e8 && io->e16 && io->ind && io->mask_c && io->mask_m && io->mask_f && io->streams && io->nbytes,
                 CGIC_ERR_INVALID, "compress_image: NULL input / output");
    CGIC_REQUIRE(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, CGIC_ERR_INVALID, "compress_image: H and W must be positive multiples of 16");
    const int64_t h = H / 4, w = W / 4;
    int rc;
    if (io->x_is_u8)
        rc = cgic_entropy_maps_u8(reinterpret_cast<const unsigned char *>(io->x), B, H, W, bins, nbins, sigma, io->x_out, io->e8, io->e16, io->flat8, stream);
    else
        rc = cgic_entropy_maps_f32(reinterpret_cast<const float *>(io->x), B, H, W, bins, nbins, sigma, io->e8, io->e16, io->flat8, stream);
    if (rc) return rc;
    // (the refinement reads the frames themselves -- ToTensor's arithmetic is part of the patch evaluation -- or the fp32 pixels)
    cgic_pixels px;
    memset(&px, 0, sizeof(px));
    px.x = io->x; px.is_u8 = io->x_is_u8 ? 1 : 0; px.bins = bins; px.nbins = nbins; px.sigma = sigma; px.flat8 = io->flat8;
    px.scratch = io->ws_refine; px.scratch_bytes = io->ws_refine_bytes;
    // (every segment is refined: in the router's workgroup up to 768x768, beyond that through patched copies in ws_refine --
    // which cgic_vq_forward_route_f32 then REQUIRES: a call without it fails instead of routing from unrefined maps)
    const cgic_pixels *refine = &px;
    int mode = 0;
    rc = cgic_vq_forward_route_f32(io->z, B, h * w, codebook, K, e_dim, beta, legacy, io->ind, io->z_q, io->loss, io->ws_vq, io->e16, io->e8,
                                   H / 16, W / 16, coarse_ratio, medium_ratio, 1, io->mask_c, io->mask_m, io->mask_f, nullptr, &mode, nullptr,
                                   prepared, refine, stream);
    if (rc) return rc;
    if (mode_out) *mode_out = mode;
    rc = cgic_compress_streams(t, io->ind, io->mask_c, io->mask_m, io->mask_f, B, h, w, mode, io->streams, io->slot, io->nbytes, io->hist,
                               io->ws_compress, stream);
    if (rc || !io->dind) return rc;
    return cgic_decompress_streams(t, io->streams, io->slot, io->nbytes, B, h, w, mode, io->dind, io->dmask_c, io->dmask_m, io->dmask_f,
                                   io->dz_q ? codebook : nullptr, K, e_dim, io->dz_q, nullptr, nullptr, io->status, io->ws_decompress, decoder,
                                   stream);
}

extern "C" int cgic_compress_tiled(const cgic_table *t, const float *codebook, int K, int e_dim, const void *prepared, const void *src,
                                   int src_is_u8, int64_t N, int64_t H, int64_t W, int ngroups, const cgic_tile_group *groups,
                                   double coarse_ratio, double medium_ratio, float beta, int legacy, const float *bins, int nbins, float sigma,
                                   int decoder, int *mode_out, cgic_stream_t stream)
{
    CGIC_REQUIRE(src && groups && ngroups >= 1 && ngroups <= kMaxGroups && N > 0, CGIC_ERR_INVALID, "compress_tiled: bad arguments (1..%d shape groups)", kMaxGroups);
    double shares[kMaxGroups];
    for (int g = 0; g < ngroups; ++g) {
        const cgic_tile_group &G = groups[g];
        const cgic_image_io &io = G.io;
        CGIC_REQUIRE(G.ntiles >= 1 && G.origins && G.th > 0 && G.tw > 0 && G.th % 16 == 0 && G.tw % 16 == 0, CGIC_ERR_INVALID, "compress_tiled: group %d: bad tile shape", g);
        CGIC_REQUIRE(io.z && io.x_out && io.e8 && io.e16 && io.flat8 && io.ind && io.mask_c && io.mask_m && io.mask_f && io.streams && io.nbytes,
                     CGIC_ERR_INVALID, "compress_tiled: group %d: NULL buffer", g);
        shares[g] = G.share;
    }
    int rc = cgic_group_begin(ngroups, shares);
    if (rc) return rc;
    int mode = cgic_router_mode(coarse_ratio, medium_ratio);
    for (int g = 0; g < ngroups && !rc; ++g) {
        const cgic_tile_group &G = groups[g];
        const cgic_image_io &io = G.io;
        const int64_t B = N * G.ntiles, h = G.th / 4, w = G.tw / 4;
        rc = cgic_group_select(g);
        if (!rc) rc = cgic_entropy_maps_tiles(src, src_is_u8, N, H, W, G.ntiles, G.origins, G.th, G.tw, bins, nbins, sigma, io.x_out, io.e8, io.e16,
                                              io.flat8, stream);
        cgic_pixels px;
        memset(&px, 0, sizeof(px));
        px.x = io.x_out; px.is_u8 = 0; px.bins = bins; px.nbins = nbins; px.sigma = sigma; px.flat8 = io.flat8;
        px.scratch = io.ws_refine; px.scratch_bytes = io.ws_refine_bytes;
        // (tiles beyond 768x768 cannot be refined inside a launch group: cgic_vq_forward_route_f32 refuses them -- never unrefined)
        const cgic_pixels *refine = &px;
        if (!rc) rc = cgic_vq_forward_route_f32(io.z, B, h * w, codebook, K, e_dim, beta, legacy, io.ind, io.z_q, io.loss, io.ws_vq, io.e16, io.e8,
                                                G.th / 16, G.tw / 16, coarse_ratio, medium_ratio, 1, io.mask_c, io.mask_m, io.mask_f, nullptr, nullptr,
                                                nullptr, prepared, refine, stream);
        if (!rc) rc = cgic_compress_streams(t, io.ind, io.mask_c, io.mask_m, io.mask_f, B, h, w, mode, io.streams, io.slot, io.nbytes, io.hist,
                                            io.ws_compress, stream);
        if (!rc && io.dind)
            rc = cgic_decompress_streams(t, io.streams, io.slot, io.nbytes, B, h, w, mode, io.dind, io.dmask_c, io.dmask_m, io.dmask_f,
                                         io.dz_q ? codebook : nullptr, K, e_dim, io.dz_q, nullptr, nullptr, io.status, io.ws_decompress, decoder,
                                         stream);
    }
    if (rc) { cgic_group_abort(); return rc; }
    if (mode_out) *mode_out = mode;
    rc = cgic_group_launch(stream);
    return rc < 0 ? rc : CGIC_OK;
}
