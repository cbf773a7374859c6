// jpt_kernels_wf2.hip -- the native (fast) route: persistent-block wavefront path tracer.
//
// All `n_frames` frames of a render are in flight at once as independent paths (pixel x frame); per-pixel
// RNG streams (main.glsl:409, :386) are carried with the path, so the schedule cannot change results.
// The grid is G persistent blocks.  Block s owns SEGMENT s of every queue: chunks of 64 consecutive path ids
// (path = pixel slot * n_frames + frame: every frame's sample of a few neighbouring pixels of an 8x8 tile)
// are dealt round-robin to segments, so each segment is a uniform sample of the image and all hand-offs
// between kernels stay inside a segment -- no global atomics anywhere on the path.
//
//   wf2_primary  bounce 0: generates primary rays (main.glsl:405-421) and traces them.  Sky misses are
//                finished on the spot; hits are packed into the segment's queue (LDS counter).
//   wf2_shade    one path vertex per queue entry (main.glsl:378-397 + brdfs.glsl); survivors' next rays are
//                packed to the front of the segment's next queue (wave ballot + prefix popcount + LDS base).
//   wf2_trace    bounces >= 1: closest hit of every queued ray.
//   wf2_finish   the few paths wf2_shade set aside (hits the reference's traversal cannot reach, exact distance ties).
//   wf2_accumulate  frames IN ORDER per pixel (progressive_rendering.glsl:33-37), display image, depth.
//   (large scenes: the tracing kernels' TAIL instantiations finish their few very long walks with a whole wave per ray,
//   coop_walk_call)
//
// Tracing kernels keep every lane busy: a lane whose ray is finished takes the next ray of the block's
// segment (cursor in LDS) while its neighbours keep walking -- the wave never waits for its slowest ray.
// Per-lane traversal stacks live in LDS (20 entries x 256 lanes, deeper levels spill to scratch).
//
// launch_wf2_render runs one render: a blocking render of many paths splits its frames into two groups that run
// this pipeline concurrently on two streams; asynchronous renders are pipelined one level up (jpt_capi.cpp).
#include <algorithm>
#include <cstdlib>

#include "jpt_trace_core.h"
#include "jpt_tie_walk.h"
#include "jpt_tuning.h"

// the primary kernel carries the ray set-up and the sky-cull test besides the walk: 74 VGPRs on its own.  Round 1 (float
// records) ran it at 6 waves/SIMD because 72 registers spilled; with the quantised records two registers are all the
// cap takes away and 7 waves/SIMD are 0.3-0.5 % ahead of 6 (four A/B repetitions, tools/ab_rates.sh); 5 are 1 % behind
#ifndef JPT_PRIMARY_WAVES
#define JPT_PRIMARY_WAVES 7
#endif

namespace jpt {

namespace {

constexpr int kBlock = kTraceBlock;
constexpr uint32_t kSegments = 256u * JPT_WAVES_PER_SIMD;  // persistent grid: one 256-thread block per CU per wave/SIMD
struct WfTune {
    int refill_idle;     // refill when at least this many lanes of a wave are idle
    int primary_refill_idle;   // ... in the primary launch
    int node_min_lanes;  // leave the node loop when fewer lanes than this still descend
    int leaf_min_lanes;  // a leaf / instance phase with fewer takers than this is put off to the next round, as long as
    int inst_min_lanes;  // other lanes of the wave can make progress meanwhile
    int phase_frac16;    // ... and never more than this many sixteenths of the wave's active rays
    int tail_rounds;     // TAIL launches: a wave enters its tail phase this many rounds after the queue ran dry ...
    int tail_lanes;      // ... once it holds at most this many rays
};

// One round of the walk for every active lane of the wave, "while-while" style so that lanes in different
// states do not serialise each other's code: (1) a tight loop of internal-record steps (lanes that reach a
// leaf wait), (2) triangle leaves, (3) instance entries.  Returns true for lanes whose walk is complete.
template <bool COUNT, bool W4>
__device__ __forceinline__ bool walk_round(Traversal<COUNT, W4>& tr, bool active, const WideSceneDev& sc,
                                           const typename Traversal<COUNT, W4>::Stack& st, DevCounters& cnt, const WfTune& tune, uint32_t& steps)
{
    const int kNodeMinLanes = tune.node_min_lanes;
    const bool lane0 = (threadIdx.x & 63) == 0;
    if (COUNT && lane0) cnt.phase[0]++;
    // A thinly attended leaf or instance phase waits for more takers while the wave has other work -- lanes that still
    // descend (the next round's record loop) or the other phase: a phase costs the wave the same whether 4 or 40 lanes take
    // part.  The thresholds shrink with the wave's rays (never more than phase_frac16 sixteenths of them): a wave that is
    // draining must not make its last rays wait.  C3 -3.5 %, close-up -5.5 %, C2 -3 % (tools/sweep_sched.sh: a plateau
    // from 8 to 32 lanes for leaves and 4 to 16 for instance entries at a quarter of the rays; without the cap small
    // renders lose 10 %).  Skipping a thinly attended record loop in the same way changes nothing.
    const int na = __popcll(__ballot(active));
    const int cap = (na * tune.phase_frac16) >> 4;
    const int lmin = tune.leaf_min_lanes < cap ? tune.leaf_min_lanes : cap;
    const int imin = tune.inst_min_lanes < cap ? tune.inst_min_lanes : cap;
    for (int it = 0; it < 64; it++) {
        if (active && !tr.have && tr.sp > 0) tr.pop_next(st);
        const bool want = active && tr.wants_node();
        const unsigned long long m = __ballot(want);
        if (m == 0 || (it > 0 && __popcll(m) < kNodeMinLanes)) break;
        if (COUNT && lane0) {
            cnt.phase[1]++;
            cnt.phase[2] += (unsigned long long)__popcll(m);
        }
        if (want) tr.node_step(sc, st, cnt);
        if (COUNT && want) steps++;
    }
    bool wl = active && tr.wants_leaf();
    bool wi = active && tr.wants_instance();
    if (W4 && (tune.leaf_min_lanes > 1 || tune.inst_min_lanes > 1)) {   // (reference trees, with leaves of up to 64 triangles, lose 5 % by waiting)
        const int nl = __popcll(__ballot(wl)), ni = __popcll(__ballot(wi));
        const bool nodes_left = __any(active && tr.wants_node()) || __any(active && !tr.have && tr.sp > 0);
        const bool run_l = nl >= lmin || !(nodes_left || ni >= imin);
        const bool run_i = ni >= imin || !(nodes_left || (run_l && nl > 0));
        wl = wl && run_l;
        wi = wi && run_i;
    }
    if (COUNT) {
        const int nl = __popcll(__ballot(wl)), ni = __popcll(__ballot(wi));
        if (lane0 && nl) {
            cnt.phase[3]++;
            cnt.phase[4] += (unsigned long long)nl;
        }
        if (lane0 && ni) {
            cnt.phase[5]++;
            cnt.phase[6] += (unsigned long long)ni;
        }
    }
    if (wl) tr.leaf_step(sc, cnt);
    if (wi) tr.instance_step(sc, st, cnt);
    if (COUNT && (wl || wi)) steps++;
    return active && tr.finished();
}

// What a render writes once and reads once -- ray and hit queues, finished paths' colours, the framebuffers -- goes past the caches
// with the non-temporal hint (`nt` on the load / store), so that 200 MB of hand-offs per launch do not push the tree's records out
// of the 4 MB L2s (C3 -2.5 %, the 4 M-triangle scene -3 %: round 5).
typedef float jpt_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t jpt_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_ld4(const float4* p)
{
    const jpt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const jpt_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_st4(float4* p, const float4 v)
{
    jpt_v4f w;
    w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<jpt_v4f*>(p));
}
__device__ __forceinline__ uint4 stream_ldu4(const uint4* p)
{
    const jpt_v4u v = __builtin_nontemporal_load(reinterpret_cast<const jpt_v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_stf(float* p, const float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ uint32_t stream_ldu(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_stu(uint32_t* p, const uint32_t v) { __builtin_nontemporal_store(v, p); }

constexpr uint32_t kHasRadiance = 0x80000000u;   // queue entry, direction.w: rad[path] holds the path's radiance (else it is 0)
constexpr uint32_t kPathMask = 0x7fffffffu;

struct Wf2Buffers {
    float4* ray_o[2];   // queue entry: origin.xyz, w = seed.y bits of the path (bounces >= 1)
    float4* ray_d[2];   // direction.xyz, w = path id bits | kHasRadiance
    float4* hit_a;      // t, u, v, tri bits          (same index as the ray)
    uint32_t* hit_b;    // inst | front << 31
    float4* thr_q[2];   // queue entry: the path's throughput.xyz, w = seed.x bits (dense, beside the ray: read with it)
    float4* thr;        // per path: where a set-aside path parks that float4 until wf2_finish (otherwise untouched)
    float4* rad;        // per path: radiance.xyz -- touched only by vertices that ADD radiance (emitters) and by paths that
                        // carry some (kHasRadiance in the queue entry): most vertices neither read nor write it
                        // (indexed by path; HDR_F32 mode: also the finished paths' output; w = seed.y of a set-aside path)
    uint32_t* fin8;     // REF_LDR8 mode: by path, the finished path's radiance as the rgba8 main.glsl:434 stores (4 bytes, not 16)
    float* first_depth; // per slot of the LAST frame: distance of the first hit (or far)
    uint32_t* qcount;   // [max_bounces + 2][kSegments] queue sizes; row b = rays traced in bounce b (b >= 1)
    uint32_t* redo_count;  // [0]: paths set aside because the reference cannot reach their hit (wf2_finish) ...
    float4* redo_rec;      // ... two float4 each: the vertex's ray, origin.w = bounce bits, direction.w = path id bits
    uint32_t redo_cap;     // records redo_rec holds (a few paths in 10^7 are set aside: not one record per path).  A hit
                           // that finds the buffer full is shaded as found -- the native tree's closest hit, without the
                           // reference's crack -- and counted in redo_count[1] (jpt_stats.set_aside_dropped)
};

// n / d for a divisor fixed per launch: one multiply-high with floor(2^32 / d) and one correction step (the
// estimate is never too large and at most one too small for n < 2^31) instead of the ~35-instruction emulated divide.
struct FastDiv {
    uint32_t d, m;
};
inline FastDiv make_fastdiv(uint32_t d) { return FastDiv{d ? d : 1u, d > 1u ? (uint32_t)(0x100000000ull / d) : 0xffffffffu}; }
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f)
{
    uint32_t q = __umulhi(n, f.m);
    const uint32_t r = n - q * f.d;
    return r >= f.d ? q + 1u : q;
}

struct Wf2Dims {
    // The paths of a render are those of the WINDOW: the tile-aligned rectangle of the image outside which every pixel is
    // sky-culled (the whole image when the cull is off).  Tiles, slots, chunks and path ids count inside it.
    int32_t tile_x0, tile_y0;  // the window's first tile (local tile rows)
    int32_t tiles_x, tiles_y;  // the window's size in tiles
    int32_t full_tiles_x, full_tiles_y;  // the context's whole share of the image in tiles (wf2_accumulate walks it)
    uint32_t tiles_per_frame;
    uint32_t slots_per_frame;  // tiles_per_frame * 64
    uint32_t n_chunks;         // tiles_per_frame * n_frames
    uint32_t seg_cap;          // entries per segment
    uint32_t run_shift;        // a segment is dealt runs of 2^run_shift consecutive chunks (neighbouring tiles of one frame)
    FastDiv by_tiles_x, by_full_tiles_x;
    FastDiv by_frames;         // path -> (slot, frame), path_slot_frame; wf2_accumulate: item -> (pixel, frame)
    int32_t acc_groups = 1;    // (wf2_accumulate: the frame groups whose blocks of rad / fin8 it reads, group_frames)
};

__device__ __forceinline__ void slot_to_pixel(uint32_t slot, const Wf2Dims& dm, int& px, int& ly)
{
    const uint32_t tile = slot >> 6, lane = slot & 63u;
    const uint32_t ty = fdiv(tile, dm.by_tiles_x), tx = tile - ty * (uint32_t)dm.tiles_x;
    const uint32_t ix = lane & 7u, iy = lane >> 3;
    px = (int)((tx + (uint32_t)dm.tile_x0) * 8u + ix);
    ly = (int)((ty + (uint32_t)dm.tile_y0) * 8u + iy);
}

// Path ids count (pixel slot, frame) pairs with the FRAME running fastest: path = slot * n_frames + frame (frames of the
// launch's group).  The primary launch deals a wave 64 consecutive ids -- every frame's sample of 64 / n_frames neighbouring
// pixels of one tile row -- so whatever is indexed by path (rad, fin8, thr) is touched in runs, by the primary launch and by
// the queues' order after it, and wf2_accumulate reads a pixel's frames as one run.
__device__ __forceinline__ void path_slot_frame(uint32_t path, const Wf2Dims& dm, uint32_t n_frames, uint32_t& slot, uint32_t& f)
{
    slot = fdiv(path, dm.by_frames);
    f = path - slot * n_frames;
}

// a path is over: its radiance goes where wf2_accumulate reads it -- in REF_LDR8 mode already as the rgba8 value of
// main.glsl:434 (the accumulation sums exactly those quantised values, progressive_rendering.glsl:33), 4 bytes per path
__device__ __forceinline__ void store_final(const Wf2Buffers& wb, int accum_mode, uint32_t p, f3 r)
{
    if (accum_mode == 0) stream_stu(&wb.fin8[p], unorm8(r.x) | (unorm8(r.y) << 8) | (unorm8(r.z) << 16));
    else stream_st4(&wb.rad[p], make_float4(r.x, r.y, r.z, 0.0f));
}

// Is pixel (px, py) outside the screen rectangles of all the boxes the TLAS root offers (SkyCull, jpt_kernels.h)?  Its
// primary ray then fails every box test of the root: sky.  Such (pixel, frame) paths are not stored anywhere: the primary
// launch skips them and wf2_accumulate recomputes their sky colour from (x, y, frame).
__device__ __forceinline__ bool sky_culled(const SkyCull& cull, int px, int py)
{
    bool outside = cull.n >= 0;
    for (int k = 0; k < 4; k++)
        if (k < cull.n && px >= cull.x0[k] && px <= cull.x1[k] && py >= cull.y0[k] && py <= cull.y1[k]) outside = false;
    return outside;
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask, int lane)
{
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// ---- a long walk, finished by a whole wave -----------------------------------------------------------------------------
//
// A launch cannot end before its longest ray does, and a lone ray advances one dependent fetch at a time (~0.8 us a step on an
// otherwise idle chip).  On the 1 M-triangle scene 40 of 16.6 M primary rays take 1 000 - 5 310 record steps (99.99 % take
// fewer than 256): the primary launch spent 4.4 of its 7.6 ms waiting for them (profiles/r04/r04g_tail_probe.txt,
// r04h_walk_hist.txt).  coop_walk walks ONE ray with all 64 lanes of a wave, again from the root: a lane walks one subtree with
// its private stack, gives the entry on top of that stack to the wave's pool (LDS) whenever lanes are idle, idle lanes take
// entries from the pool (an entry carries its level: TLAS, or the instance it belongs to), and after every round all lanes adopt
// the closest distance any of them has found, so every lane culls with it.  The closest hit does not depend on the order of the
// tests; two lanes that both keep a triangle at the final distance are an exact tie and flagged as one (kHitTied), like two such
// triangles met by one walk.  5 310 dependent steps become ~150 rounds: S-unique's primary launch 8.9 -> 4.2 ms (profiles/r04/
// r04l_coop_ab.txt).  Used as the TAIL phase of the tracing launches on scenes of >= 200 000 triangles: on the small scenes no ray
// is long and the launches' tails are the drain of ordinary rays.  (Round 4's other form -- the long rays listed and walked by a
// follow-up launch, wf2_long -- cost small renders 15 % in empty launches and was removed in round 6: LAB_NOTEBOOK.md.)
constexpr int kCoopPool = 64;      // pool entries per wave (two words each)

__device__ __forceinline__ float wave_min_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin_(v, __shfl_xor(v, off));
    return v;
}

// (`pool`: 2 * kCoopPool words of LDS, word w at pool[(w >> 6) * pool_stride + (w & 63)] -- a flat array with pool_stride = 64,
// or two rows of the block's stack array, see the tail phase of the tracing kernels)
template <bool COUNT>
__device__ __forceinline__ TraceHit coop_walk(const WideSceneDev& sc, const typename Traversal<COUNT, true>::Stack st, int32_t* __restrict__ pool,
                                           f3 ro, f3 rd, DevCounters& cnt, const uint32_t pool_stride = 64u)
{
    using Walk = Traversal<COUNT, true>;
    const int lane = threadIdx.x & 63;
    auto pw = [&](uint32_t w) -> int32_t& { return pool[(w >> 6) * pool_stride + (w & 63u)]; };
    constexpr uint32_t kNone = 0xffffffffu;   // hit.tri of a lane that holds no triangle of its own at hit.t
    Walk tr;
    tr.begin(sc, ro, rd);
    tr.hit.tri = kNone;
    bool active = lane == 0 && tr.have;
    if (lane != 0) tr.have = false;
    uint32_t n_pool = 0;
    uint32_t my_steps = 0;   // (counting builds: record steps of this lane; their sum is the walk's length, jpt_stats.walk_steps_*)
    for (;;) {
        // idle lanes take pool entries (newest first)
        {
            const unsigned long long idle = __ballot(!active);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t take = n_pool < n_idle ? n_pool : n_idle;
            if (take) {
                const uint32_t r = lanes_below(idle, lane);
                if (!active && r < take) {
                    const int32_t ref = pw(2u * (n_pool - 1u - r));
                    const uint32_t ctx = (uint32_t)pw(2u * (n_pool - 1u - r) + 1u);
                    tr.sp = 0;
                    tr.cur = ref;
                    tr.have = true;
                    tr.in_blas = (ctx & 1u) != 0u;
                    tr.cur_inst = ctx >> 1;
                    if (tr.in_blas) {   // the instance's local ray (instance_step's arithmetic, main.glsl:319-320)
                        const WideInstance* ip = sc.instances + tr.cur_inst;
                        const float4 m0 = ld4(&ip->inv[0]), m1 = ld4(&ip->inv[4]), m2 = ld4(&ip->inv[8]);
                        tr.o = mk3(m0.x * ro.x + m0.w * ro.y + m1.z * ro.z + m2.y, m0.y * ro.x + m1.x * ro.y + m1.w * ro.z + m2.z,
                                   m0.z * ro.x + m1.y * ro.y + m2.x * ro.z + m2.w);
                        tr.d = mk3(m0.x * rd.x + m0.w * rd.y + m1.z * rd.z, m0.y * rd.x + m1.x * rd.y + m1.w * rd.z,
                                   m0.z * rd.x + m1.y * rd.y + m2.x * rd.z);
                    } else {
                        tr.o = ro;
                        tr.d = rd;
                    }
                    tr.set_level();
                    active = true;
                }
                n_pool -= take;
            }
        }
        if (!__any(active)) break;   // (and the pool is empty: an idle lane would have taken from it)
        if (COUNT && active && (tr.have || tr.sp > 0)) my_steps++;
        if (active && !tr.step(sc, st, cnt)) active = false;
        // every lane culls with the closest distance found anywhere; a lane whose own triangle is farther no longer holds one
        {
            const float t = wave_min_f(tr.hit.t);
            if (t < tr.hit.t) {
                tr.hit.t = t;
                tr.hit.tri = kNone;
                tr.hit.inst = 0u;
            }
        }
        // while lanes are idle, lanes with pending entries give the one on top of their stack to the pool (never the sentinel
        // that marks "leave the instance": what lies above it belongs to the instance the lane is in)
        {
            const uint32_t n_idle = (uint32_t)__popcll(__ballot(!active));
            const uint32_t want = n_idle > n_pool ? n_idle - n_pool : 0u;
            if (want) {
                int32_t top = kSentinel;
                const bool may = active && tr.sp > 0;
                if (may) {   // peek
                    const int row = tr.sp - 1;
                    top = row < st.lds_entries ? st.lds[row * st.stride] : (row < st.lds_entries + st.spill_entries ? st.spill[row - st.lds_entries] : kSentinel);
                }
                const bool give = may && top != kSentinel;
                const unsigned long long gm = __ballot(give);
                const uint32_t r = lanes_below(gm, lane);
                const uint32_t room = (uint32_t)kCoopPool - n_pool;
                const uint32_t lim = want < room ? want : room;
                if (give && r < lim) {
                    tr.sp--;
                    pw(2u * (n_pool + r)) = top;
                    pw(2u * (n_pool + r) + 1u) = (int32_t)((tr.in_blas ? 1u : 0u) | (tr.cur_inst << 1));
                }
                const uint32_t given = (uint32_t)__popcll(gm);
                n_pool += given < lim ? given : lim;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) my_steps += (uint32_t)__shfl_xor((int)my_steps, off);
        if (lane == 0) count_walk(cnt, my_steps);
    }
    // the closest hit: the lanes that hold a triangle at the final distance
    const float best = wave_min_f(tr.hit.t);
    const unsigned long long win = __ballot(tr.hit.tri != kNone && tr.hit.t == best);
    TraceHit out;
    out.t = 1e9f;
    out.u = out.v = 0.0f;
    out.tri = out.inst = 0u;
    out.front = false;
    if (win && best < 1e9f) {
        // Several lanes holding a triangle at the final distance: WHICH lanes walked which subtrees depends on the pool's timing, so
        // the winner must not be "the lowest lane" -- without the reference's trees to re-decide the tie (watertight scenes,
        // uploads that are not in pre-order) the flagged winner is shaded as it is, and the image would depend on the schedule
        // (ADVICE r04).  The winner is a function of the ray alone: the largest triangle index, then the largest instance.
        unsigned long long pick = win;
        if (__popcll(win) > 1) {
            const bool mine = ((win >> lane) & 1ull) != 0ull;
            uint32_t kt = mine ? tr.hit.tri : 0u;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)kt, off);
                kt = o > kt ? o : kt;
            }
            const bool top = mine && tr.hit.tri == kt;
            uint32_t ki = top ? ((tr.hit.inst >> kInstBits) & kInstMask) : 0u;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)ki, off);
                ki = o > ki ? o : ki;
            }
            pick = __ballot(top && ((tr.hit.inst >> kInstBits) & kInstMask) == ki);
        }
        const int w = __ffsll((long long)pick) - 1;
        out.t = best;
        out.u = __shfl(tr.hit.u, w);
        out.v = __shfl(tr.hit.v, w);
        out.tri = (uint32_t)__shfl((int)tr.hit.tri, w);
        out.front = __shfl((int)tr.hit.front, w) != 0;
        const uint32_t wi = (uint32_t)__shfl((int)tr.hit.inst, w);
        const uint32_t found_in = (wi >> kInstBits) & kInstMask;
        const bool tied = __popcll(win) > 1 || (wi & kHitTied) != 0u;
        out.inst = found_in | (found_in << kInstBits) | (tied ? kHitTied : 0u);
    }
    return out;
}

// The walk as the TAIL of a tracing launch: a wave whose block's queue has been dry for
// `tail_rounds` rounds and that is down to `tail_lanes` rays stops walking them lane by lane -- a lone ray advances one dependent
// step per microsecond, and a small render's launches each last as long as their longest walk (tools/step_latency.py: 64 x 64
// pixels, longest walk 42 steps, launch 48 us; 1920 x 1080 x 1: 102 steps, 110 us) -- and walks them one after the other with
// all its lanes, again from the root.  The call is OUT OF LINE so that the walk's registers are not the hot loop's (inlined
// there it cost the loop spills and every launch 10-20 %, round 4); its pool is the top two rows of the wave's own columns of
// the block's stack array, its private stacks the rows below.
template <bool COUNT>
__device__ __attribute__((noinline)) void coop_walk_call(const WideSceneDev* __restrict__ scp, int32_t* lds_col, int32_t* pool, float ox, float oy,
                                                         float oz, float dx, float dy, float dz, DevCounters* cntp, TraceHit* out)
{
    int32_t spill[kStackSpill + 2];   // (the two LDS rows the pool takes are made up for here)
    const typename Traversal<COUNT, true>::Stack st{lds_col, spill, kTraceBlock, kStackLds - 2, kStackSpill + 2};
    const WideSceneDev sc = *scp;
    DevCounters none = {};
    *out = coop_walk<COUNT>(sc, st, pool, mk3(ox, oy, oz), mk3(dx, dy, dz), COUNT ? *cntp : none, (uint32_t)kTraceBlock);
}

// ---- bounce 0: generate + trace ------------------------------------------------------------------------

template <bool COUNT, bool W4, bool TAIL = false>   // TAIL: the wave finishes its last, long walks itself, all lanes on one ray (coop_walk_call)
__global__ __launch_bounds__(kBlock, JPT_PRIMARY_WAVES) void wf2_primary(WideSceneDev sc, Wf2Buffers wb, Wf2Dims dm, FrameParams fp, RefCamera cam,
                                                      WfTune tune, SkyCull cull, DevCounters* __restrict__ counters)
{
    __shared__ int32_t stack[kStackLds * kBlock];
    __shared__ uint32_t s_cursor, s_out;
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x;
    // runs seg, seg + G, seg + 2G, ... of 2^run_shift consecutive chunks belong to this block
    const uint32_t run_mask = (1u << dm.run_shift) - 1u;
    const uint32_t n_runs = (dm.n_chunks + run_mask) >> dm.run_shift;
    const uint32_t my_runs = seg < n_runs ? (n_runs - seg + kSegments - 1u) / kSegments : 0u;
    const uint32_t n = (my_runs << dm.run_shift) * 64u;  // (the last run of the image may be short: checked per entry)
    if (threadIdx.x == 0) {
        s_cursor = 0;
        s_out = 0;
    }
    // The queue sizes of bounces >= 1 are accumulated with atomics by wf2_shade and must start from zero: block s clears segment s's
    // word of every later row (and block 0 the set-aside counts behind them) -- this launch completes before the first wf2_shade
    // starts, and one launch fewer per render is one link less in a chain of a dozen (a memset used to do this).
    if ((int)threadIdx.x >= 1 && (int)threadIdx.x <= fp.max_bounces + 1) wb.qcount[(size_t)threadIdx.x * kSegments + seg] = 0u;
    if (seg == 0 && threadIdx.x >= 192u) wb.redo_count[threadIdx.x - 192u] = 0u;
    __syncthreads();
    int32_t spill[kStackSpill];
    const typename Traversal<COUNT, W4>::Stack my_stack{&stack[threadIdx.x], spill, kTraceBlock, kStackLds};
    const size_t seg_base = (size_t)seg * dm.seg_cap;
    uint32_t dry_rounds = 0;   // (TAIL: rounds of this wave since the block's queue ran dry; wave-uniform)
    DevCounters cnt = {};
    Traversal<COUNT, W4> tr;
    bool active = false, exhausted = false;
    bool unsaved = false;   // this lane's finished walk has not left its result yet (see wf2_trace: written when the wave refills)
    uint32_t path = 0;
    uint32_t walk_steps = 0;   // (counting builds: record steps of this lane's ray)
    auto save_results = [&]() {
        // hits are packed into the segment's bounce-0 queue (main.glsl:349), one counter update per wave
        const bool is_hit = unsaved && tr.hit.t < 1e9f;
        const unsigned long long hm = __ballot(is_hit);
        if (hm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_out, (uint32_t)__popcll(hm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (is_hit) {
                const uint32_t j = base + lanes_below(hm, lane);
                stream_st4(&wb.ray_o[0][seg_base + j], make_float4(tr.wo.x, tr.wo.y, tr.wo.z, 0.0f));
                stream_st4(&wb.ray_d[0][seg_base + j], make_float4(tr.wd.x, tr.wd.y, tr.wd.z, __uint_as_float(path)));
                stream_st4(&wb.hit_a[seg_base + j], make_float4(tr.hit.t, tr.hit.u, tr.hit.v, __uint_as_float(tr.hit.tri)));
                stream_stu(&wb.hit_b[seg_base + j], tr.hit.inst | (tr.hit.front ? 0x80000000u : 0u));
            }
        }
        if (unsaved && !is_hit) {   // sky: radiance += 1 * sampleSky(d), path over (main.glsl:380,395-397)
            uint32_t pslot, f;
            path_slot_frame(path, dm, (uint32_t)fp.n_frames, pslot, f);
            const f3 sky = mk3(0.0f, 0.0f, 0.0f) + mk3(1.0f, 1.0f, 1.0f) * sample_sky(tr.wd);
            store_final(wb, fp.accum_mode, path, sky);
            if ((int)f == fp.depth_frame) wb.first_depth[pslot] = cam.far_;
        }
        unsaved = false;
    };

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= tune.primary_refill_idle) {
            if (__any(unsaved)) save_results();
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(&s_cursor, (uint32_t)n_idle);
            start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
            if (start + (uint32_t)n_idle >= n) exhausted = true;
            if (start < n && !active) {
                const uint32_t idx = start + lanes_below(idle, lane);
                if (idx < n) {
                    const uint32_t j = idx >> 6;  // local chunk number: run j >> run_shift, position j & run_mask
                    const uint32_t chunk = ((seg + (j >> dm.run_shift) * kSegments) << dm.run_shift) + (j & run_mask);
                    // a wave takes 64 consecutive path ids: chunk c is (tile c / F, part c % F), its lanes the samples part * 64 + lane of that
                    // tile, sample s being frame s % F of the tile's pixel s / F
                    uint32_t slot = 0, f = 0;
                    if (chunk < dm.n_chunks) path_slot_frame(chunk * 64u + (idx & 63u), dm, (uint32_t)fp.n_frames, slot, f);
                    int px, ly;
                    slot_to_pixel(slot, dm, px, ly);
                    if (chunk < dm.n_chunks && px < fp.width && ly < fp.local_rows) {
                        const int py = local_to_global_row(ly, fp);
                        path = slot * (uint32_t)fp.n_frames + f;
                        if (COUNT) cnt.rays++;
                        // a pixel outside the screen rectangles of all the boxes the TLAS root offers: the walk would
                        // expand the root, fail every box test and end in the sky -- nothing is generated, traced or
                        // stored for it here; wf2_accumulate makes up its colour from (x, y, frame)
                        if (sky_culled(cull, px, py)) {
                            if (COUNT) {
                                cnt.tlas_expand++;
                                cnt.phase[7]++;
                            }
                        } else {
                            uint32_t sx, sy;
                            const Ray ray = primary_ray(cam, fp.width, fp.height, px, py, fp.frame_index + f, sx, sy);
                            tr.begin(sc, ray.o, ray.d);
                            active = true;
                            walk_steps = 0;
                        }
                    }
                }
            }
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        {
            if (walk_round<COUNT, W4>(tr, active, sc, my_stack, cnt, tune, walk_steps)) {
                active = false;
                unsaved = true;
                if (COUNT) count_walk(cnt, walk_steps);
            }
            if constexpr (TAIL && W4) {
                if (exhausted && ++dry_rounds >= (uint32_t)tune.tail_rounds && __popcll(__ballot(active)) <= tune.tail_lanes) break;
            }
        }
    }
    if constexpr (TAIL && W4) {
        // tail phase (coop_walk_call): the few rays this wave still holds, each walked by the whole wave
        unsigned long long left = __ballot(active);
        if (left) {
            if (__any(unsaved)) save_results();   // (the finished lanes' hits leave the registers the call is free to use)
            const int wave_col = (int)(threadIdx.x & ~63u);
            while (left) {
                const int src = __ffsll((long long)left) - 1;
                left &= left - 1ull;
                TraceHit h;
                const WideSceneDev sc_tail = sc;   // (a copy of its own: `sc` itself must not have its address taken, or the hot loop reads it from scratch)
                coop_walk_call<COUNT>(&sc_tail, &stack[threadIdx.x], &stack[(kStackLds - 2) * kTraceBlock + wave_col], __shfl(tr.wo.x, src),
                                      __shfl(tr.wo.y, src), __shfl(tr.wo.z, src), __shfl(tr.wd.x, src), __shfl(tr.wd.y, src), __shfl(tr.wd.z, src),
                                      COUNT ? &cnt : nullptr, &h);
                if (lane == src) {
                    tr.hit = h;
                    active = false;
                    unsaved = true;
                }
            }
        }
    }
    if (__any(unsaved)) save_results();
    __syncthreads();
    if (threadIdx.x == 0) wb.qcount[0 * kSegments + seg] = s_out;
    if (COUNT) flush_counters(cnt, counters);
}

// ---- bounces >= 1: trace the segment's ray queue ----------------------------------------------------------

// `chain` consecutive segments are one block's queue (entries of segment seg0, then seg0 + 1, ...): the grid is
// kSegments / chain blocks.  A deeper queue per block keeps the lanes refilled for a larger share of the launch (at
// bounce 4 of C3 a segment holds 1.7 rays per lane); the slots this leaves free are used by the launches of the
// other renders in flight (render pipelining), so chain > 1 only pays when renders are queued.
#ifndef JPT_MAX_CHAIN
#define JPT_MAX_CHAIN 4
#endif
constexpr int kMaxChain = JPT_MAX_CHAIN;
// The walk of one block's ray queue -- `n` entries in the consecutive segments seg0, seg0 + 1, .. (end[k] = entries in the first k + 1
// of them) -- with the block's stack columns and queue cursor in LDS: the body of wf2_trace.
template <bool COUNT, bool W4, bool TAIL>
__device__ __forceinline__ void trace_queue(const WideSceneDev& sc, const Wf2Buffers& wb, const Wf2Dims& dm, const int bounce, const WfTune& tune,
                                            const uint32_t seg0, const uint32_t (&end)[kMaxChain], const uint32_t n, int32_t* __restrict__ stack,
                                            uint32_t* __restrict__ s_cursor, DevCounters& cnt)
{
    const int lane = threadIdx.x & 63;
    int32_t spill[kStackSpill];
    const typename Traversal<COUNT, W4>::Stack my_stack{&stack[threadIdx.x], spill, kTraceBlock, kStackLds};
    uint32_t dry_rounds = 0;   // (TAIL: rounds of this wave since the block's queue ran dry; wave-uniform)
    const float4* __restrict__ qo = wb.ray_o[bounce & 1];
    const float4* __restrict__ qd = wb.ray_d[bounce & 1];
    Traversal<COUNT, W4> tr;
    bool active = false, exhausted = false;
    bool unsaved = false;   // this lane's finished walk has not written its hit yet
    size_t my_loc = 0;
    uint32_t walk_steps = 0;   // (counting builds: record steps of this lane's ray)
    // A finished walk's hit stays in the lane's registers until the lane takes its next ray: the hits are written when the
    // wave refills (two dozen lanes at once) instead of in the round each walk happens to end in (some lane does in
    // nearly every round: a dozen instructions per round for one or two lanes' stores).
    auto save_hit = [&]() {
        stream_st4(&wb.hit_a[my_loc], make_float4(tr.hit.t, tr.hit.u, tr.hit.v, __uint_as_float(tr.hit.tri)));
        stream_stu(&wb.hit_b[my_loc], tr.hit.inst | (tr.hit.front ? 0x80000000u : 0u));
    };

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= tune.refill_idle) {
            if (unsaved) {
                save_hit();
                unsaved = false;
            }
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(s_cursor, (uint32_t)n_idle);
            start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
            if (start + (uint32_t)n_idle >= n) exhausted = true;
            if (start < n && !active) {
                const uint32_t idx = start + lanes_below(idle, lane);
                if (idx < n) {
                    // which of the chained segments holds entry idx
                    uint32_t k = 0, first = 0;
                    for (int j = 0; j < kMaxChain - 1; j++)
                        if (idx >= end[j]) {
                            k = (uint32_t)j + 1u;
                            first = end[j];
                        }
                    const size_t loc = (size_t)(seg0 + k) * dm.seg_cap + (idx - first);
                    const float4 ro = stream_ld4(&qo[loc]), rd = stream_ld4(&qd[loc]);
                    tr.begin(sc, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z));
                    my_loc = loc;
                    active = true;
                    walk_steps = 0;
                }
            }
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        {
            if (walk_round<COUNT, W4>(tr, active, sc, my_stack, cnt, tune, walk_steps)) {
                active = false;
                unsaved = true;
                if (COUNT) count_walk(cnt, walk_steps);
            }
            if constexpr (TAIL && W4) {
                if (exhausted && ++dry_rounds >= (uint32_t)tune.tail_rounds && __popcll(__ballot(active)) <= tune.tail_lanes) break;
            }
        }
    }
    if constexpr (TAIL && W4) {
        // tail phase (see wf2_primary)
        unsigned long long left = __ballot(active);
        if (left) {
            if (unsaved) {
                save_hit();
                unsaved = false;
            }
            const int wave_col = (int)(threadIdx.x & ~63u);
            while (left) {
                const int src = __ffsll((long long)left) - 1;
                left &= left - 1ull;
                TraceHit h;
                const WideSceneDev sc_tail = sc;   // (a copy of its own: `sc` itself must not have its address taken, or the hot loop reads it from scratch)
                coop_walk_call<COUNT>(&sc_tail, &stack[threadIdx.x], &stack[(kStackLds - 2) * kTraceBlock + wave_col], __shfl(tr.wo.x, src),
                                      __shfl(tr.wo.y, src), __shfl(tr.wo.z, src), __shfl(tr.wd.x, src), __shfl(tr.wd.y, src), __shfl(tr.wd.z, src),
                                      COUNT ? &cnt : nullptr, &h);
                if (lane == src) {
                    tr.hit = h;
                    active = false;
                    save_hit();
                }
            }
        }
    }
    if (unsaved) save_hit();
}

template <bool COUNT, bool W4, bool TAIL = false>
__global__ __launch_bounds__(kBlock, JPT_WAVES_PER_SIMD) void wf2_trace(WideSceneDev sc, Wf2Buffers wb, Wf2Dims dm, int bounce, WfTune tune,
                                                    int chain, DevCounters* __restrict__ counters)
{
    __shared__ int32_t stack[kStackLds * kBlock];
    __shared__ uint32_t s_cursor;
    const uint32_t seg0 = blockIdx.x * (uint32_t)chain;
    // end[k] = entries in segments seg0 .. seg0 + k (wave-uniform, kept in scalars)
    uint32_t end[kMaxChain];
    uint32_t n = 0;
    for (int k = 0; k < kMaxChain; k++) {
        if (k < chain && seg0 + (uint32_t)k < kSegments) n += wb.qcount[(size_t)bounce * kSegments + seg0 + (uint32_t)k];
        end[k] = n;
    }
    if (n == 0) return;
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    DevCounters cnt = {};
    trace_queue<COUNT, W4, TAIL>(sc, wb, dm, bounce, tune, seg0, end, n, stack, &s_cursor, cnt);
    if (COUNT) flush_counters(cnt, counters);
}

// ---- shading: one path vertex per queue entry (main.glsl:378-397) -------------------------------------------

#ifndef JPT_SHADE_WAVES
#define JPT_SHADE_WAVES 5   // 96 VGPRs: two fewer than the body wants, five waves per SIMD instead of four for a kernel that waits on gathers (C3 +4 %); 6 spills
#endif
// One path vertex (main.glsl:378-397): the queue entry's ray and closest hit in, radiance / throughput / seeds of the path
// updated, returns true when the path goes on (no / nd = its next ray).  With reach records (JPT_BUILD_SAH) and
// check_reach, a hit the reference's own traversal could not have reached -- the world ray fails the world box the
// reference gives the instance, or the local ray fails the box of the triangle's reference leaf (jpt_types.h) -- is
// not shaded: the vertex is set aside (redo_rec; its state parked in the path's thr / rad words), `unreachable` comes back
// true, and the path leaves the wavefront to be finished by wf2_finish.
// LAST: the vertex is known to be the path's last (bounce == max_bounces): emission or sky is added and the path ends -- no BRDF
// sample, no next ray -- so the instantiation carries none of that code (wf2_shade's final launch).
template <bool COUNT, bool LAST = false, int TEX = 3>
__device__ __forceinline__ bool shade_entry(const SceneShading& sh, const Wf2Buffers& wb, const Wf2Dims& dm, const FrameParams& fp, float cam_far,
                                            int bounce, const float4 ro, const float4 rd, const float4 tin, const float4 ha, const uint32_t hb,
                                            bool check_reach, bool& unreachable, float4& no, float4& nd, float4& nt, DevCounters& cnt)
{
    unreachable = false;
    const uint32_t p = __float_as_uint(rd.w) & kPathMask;
    const bool had_radiance = (__float_as_uint(rd.w) & kHasRadiance) != 0u;
    f3 throughput, radiance;
    uint32_t sx, sy;
    uint32_t slot, f;
    path_slot_frame(p, dm, (uint32_t)fp.n_frames, slot, f);
    Ray ray;
    ray.o = mk3(ro.x, ro.y, ro.z);
    ray.d = mk3(rd.x, rd.y, rd.z);
    const bool is_hit = ha.x < 1e9f;  // main.glsl:349
    // The kernel waits on gathers, so everything whose address is known once the queue entry is here is asked for at
    // once, ahead of the branches that use it: the path's state now, the instance and the reach boxes below.
    const float4 t4 = tin;   // (the entry's throughput and seed.x: came with the ray)
    float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (bounce > 0 && had_radiance) r4 = wb.rad[p];
    Hit h;
    ShadeTriRegs stri;
    uint32_t found_in = 0;
    if (is_hit) {
        h.t = ha.x;
        h.u = ha.y;
        h.v = ha.z;
        h.tri = __float_as_uint(ha.w);
        h.inst = hb & kInstMask;                   // hitInfo.blas
        found_in = (hb >> kInstBits) & kInstMask;  // the instance whose local ray found the triangle
        const bool check = check_reach && sh.reach_tri != nullptr;
        stri = load_shade_tri(sh, h.tri);
        float4 ta, tb, ia, ib;
        if (check) {   // (whole records, unconditionally: a flag test first would put two round trips in a row)
            ta = ld4(&sh.reach_tri[h.tri].lo[0]);
            tb = ld4(&sh.reach_tri[h.tri].hi[0]);
            if (sh.n_instances > 1u) {
                ia = ld4(&sh.reach_inst[found_in].lo[0]);
                ib = ld4(&sh.reach_inst[found_in].hi[0]);
            }
        }
        // the hit instance's local ray: the expression ray_trace_tlas evaluates (main.glsl:319-320)
        const RefInstance& b = sh.instances[found_in];
        h.lo = xform_point(b.inverse_transform, ray.o);
        h.ld = xform_dir(b.inverse_transform, ray.d);
        if (check) {
            bool reached = __float_as_uint(ta.w) != 0u || slab(h.lo, rcp3(h.ld), ta.x, ta.y, ta.z, tb.x, tb.y, tb.z) < 1e30f;
            if (reached && sh.n_instances > 1u) reached = slab(ray.o, rcp3(ray.d), ia.x, ia.y, ia.z, ib.x, ib.y, ib.z) < 1e30f;
            // (an exact distance tie is as undecidable on the native tree as a crack: with the reference's trees at hand
            // it is decided where the reference decides it)
            if (sh.retrace_ties && (hb & kHitTied) != 0u) reached = false;
            if (!reached) {
                // a few paths in 10^7: the path leaves the wavefront here and is finished, exactly, by wf2_finish -- unless
                // the set-aside buffer is full (pathological scenes): then the hit is shaded as found, and counted
                const uint32_t k = atomicAdd(&wb.redo_count[0], 1u);
                if (k < wb.redo_cap) {
                    float4 rd2 = rd;
                    if (bounce > 0) {   // its seed.y (the entry's origin.w) waits in rad[path].w, beside the radiance so far
                        wb.rad[p] = make_float4(r4.x, r4.y, r4.z, ro.w);
                        wb.thr[p] = tin;
                        rd2.w = __uint_as_float(p | kHasRadiance);
                    }
                    wb.redo_rec[2 * (size_t)k] = make_float4(ro.x, ro.y, ro.z, __uint_as_float((uint32_t)bounce));
                    wb.redo_rec[2 * (size_t)k + 1] = rd2;
                    unreachable = true;
                    return false;
                }
                atomicAdd(&wb.redo_count[1], 1u);
            }
        }
    }
    if (bounce == 0) {
        // fresh path: the seed after the jitter draw (main.glsl:409-411), recomputed from (x, y, frame)
        int px, ly;
        slot_to_pixel(slot, dm, px, ly);
        prng_seed((uint32_t)px, (uint32_t)local_to_global_row(ly, fp), fp.frame_index + f, sx, sy);
        float r0, r1;
        pcg2d(sx, sy, r0, r1);
        throughput = mk3(1.0f, 1.0f, 1.0f);
        radiance = mk3(0.0f, 0.0f, 0.0f);
    } else {
        throughput = mk3(t4.x, t4.y, t4.z);
        radiance = mk3(r4.x, r4.y, r4.z);
        sx = __float_as_uint(t4.w);
        sy = __float_as_uint(ro.w);
    }
    const f3 radiance_in = radiance;
    if (COUNT && bounce > 0) cnt.rays++;
    bool alive = false;
    if (!is_hit) {
        radiance = radiance + throughput * sample_sky(ray.d);
        if (bounce == 0 && (int)f == fp.depth_frame) wb.first_depth[slot] = cam_far;  // (only a redone primary hit can turn into a miss here)
    } else {
        if (COUNT) cnt.shaded_hits++;
        const Shading s = get_shading_data<TEX>(sh, h, (hb >> 31) != 0u, stri);
        radiance = radiance + throughput * s.emission;
        if (bounce == 0 && (int)f == fp.depth_frame) wb.first_depth[slot] = length3(s.position - ray.o);
        if (!LAST && bounce < fp.max_bounces) alive = bounce_step(s, sx, sy, ray, throughput);
        if (COUNT && alive && throughput.x == 0.0f && throughput.y == 0.0f && throughput.z == 0.0f) cnt.zero_thr++;
    }
    if (alive) {
        // (radiance starts as +0 and +0 + x is x or +0, never -0: "unchanged and never written" means exactly +0)
        const bool changed = radiance.x != radiance_in.x || radiance.y != radiance_in.y || radiance.z != radiance_in.z;
        if (changed) wb.rad[p] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
        nt = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(sx));
        no = make_float4(ray.o.x, ray.o.y, ray.o.z, __uint_as_float(sy));
        nd = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(p | ((had_radiance || changed) ? kHasRadiance : 0u)));
    } else {
        store_final(wb, fp.accum_mode, p, radiance);
    }
    return alive;
}

#ifndef JPT_SHADE_NOTEX_WAVES
#define JPT_SHADE_NOTEX_WAVES 7   // without the sampler code the body takes 67 VGPRs by itself
#endif
template <bool COUNT, bool LAST = false, int TEX = 3>
__global__ __launch_bounds__(kBlock, LAST ? 8 : (TEX == 0 ? JPT_SHADE_NOTEX_WAVES : JPT_SHADE_WAVES)) void wf2_shade(SceneShading sh, Wf2Buffers wb, Wf2Dims dm, FrameParams fp, float cam_far, int bounce,
                                                    DevCounters* __restrict__ counters)
{
    // grid = (chunks per segment, segments): every 256-entry chunk of every segment is its own block, so the
    // launch is balanced however unevenly the segments are filled; blocks past a segment's end exit at once.
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.y;
    const uint32_t n = wb.qcount[(size_t)bounce * kSegments + seg];
    const uint32_t base = blockIdx.x * kBlock;
    if (base >= n) return;
    const size_t seg_base = (size_t)seg * dm.seg_cap;
    const int in = bounce & 1, out = (bounce + 1) & 1;
    DevCounters cnt = {};
    const uint32_t i = base + threadIdx.x;
    bool alive = false;
    float4 no, nd, nt;
    if (i < n) {
        const float4 ro = stream_ld4(&wb.ray_o[in][seg_base + i]), rd = stream_ld4(&wb.ray_d[in][seg_base + i]);
        const float4 tin = bounce > 0 ? stream_ld4(&wb.thr_q[in][seg_base + i]) : make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        const float4 ha = stream_ld4(&wb.hit_a[seg_base + i]);
        const uint32_t hb = stream_ldu(&wb.hit_b[seg_base + i]);
        bool unreachable;   // (set aside inside shade_entry: nothing more to do here)
        alive = shade_entry<COUNT, LAST, TEX>(sh, wb, dm, fp, cam_far, bounce, ro, rd, tin, ha, hb, true, unreachable, no, nd, nt, cnt);
    }
    if (LAST) {   // (no path goes on: nothing to pack)
        if (COUNT) flush_counters(cnt, counters);
        return;
    }
    // active-ray packing: wave ballot + prefix popcount, one atomic per wave on the SEGMENT's counter (1792
    // different words: no hot address); the order inside the next queue is irrelevant
    const unsigned long long m = __ballot(alive);
    if (m) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&wb.qcount[(size_t)(bounce + 1) * kSegments + seg], (uint32_t)__popcll(m));
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (alive) {
            const size_t j = seg_base + wbase + lanes_below(m, lane);
            stream_st4(&wb.ray_o[out][j], no);
            stream_st4(&wb.ray_d[out][j], nd);
            stream_st4(&wb.thr_q[out][j], nt);
        }
    }
    if (COUNT) flush_counters(cnt, counters);
}

// The paths wf2_shade set aside: their hit is one the reference's traversal cannot reach.  Each is finished here, after
// the render's last bounce launch and before the accumulation: from the vertex where it left the wavefront the path is
// traced with the reach tests applied to every instance entry and every accepted triangle (Traversal<.., REACH = true>:
// the closest hit among the triangles the reference can reach, i.e. the reference's answer) and shaded bounce after
// bounce by the same shade_entry, its state passing through the path's own thr / rad words.  A handful of paths per
// render, so one small grid of single-wave blocks with the whole stack in scratch; one launch per render.
template <bool COUNT, bool W4, bool EXACT>
__global__ __launch_bounds__(64) void wf2_finish(WideSceneDev sc, TieShadowDev sx, SceneShading sh, Wf2Buffers wb, Wf2Dims dm, FrameParams fp, float cam_far,
                                                 DevCounters* __restrict__ counters)
{
    const uint32_t n = wb.redo_count[0] < wb.redo_cap ? wb.redo_count[0] : wb.redo_cap;
    if (blockIdx.x >= n) return;
    constexpr int kDepth = kStackLds + kStackSpill;
    int32_t stack_mem[kDepth];   // (the whole stack in scratch; in LDS instead: no difference, profiles/r03/r03z_finish_probe.txt)
    // Every segment of a set-aside path is traced on the native tree with the two reach tests applied inside the walk: the
    // closest hit among the triangles the reference can reach.  EXACT (`sx` = the reference's own trees): when that walk
    // met a second reachable triangle at exactly the closest distance, the tie is decided where the reference decides it
    // (jpt_tie_walk.h).
    using Walk = Traversal<COUNT, W4, true>;
    const typename Walk::Stack st{nullptr, stack_mem, 0, 0, kDepth};
    DevCounters cnt = {};
    // (records are dealt to the BLOCKS first -- record k to block k % grid, lane k / grid: a handful of set-aside paths run as
    // one lane each of as many waves, side by side, instead of as divergent lanes of one wave, one after another)
    for (uint32_t k = threadIdx.x * gridDim.x + blockIdx.x; k < n; k += gridDim.x * 64u) {
        float4 ro = wb.redo_rec[2 * (size_t)k], rd = wb.redo_rec[2 * (size_t)k + 1];
        const int first = (int)__float_as_uint(ro.w);
        float4 tin = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        if (first > 0) {   // the seed.y and the throughput parked when the path was set aside
            ro.w = wb.rad[__float_as_uint(rd.w) & kPathMask].w;
            tin = wb.thr[__float_as_uint(rd.w) & kPathMask];
        }
        for (int bounce = first;; bounce++) {
            const f3 o3 = mk3(ro.x, ro.y, ro.z), d3 = mk3(rd.x, rd.y, rd.z);
            Walk tr;
            tr.begin(sc, o3, d3);
            while (tr.step(sc, st, cnt)) {
            }
            TraceHit hit = tr.hit;
            if (EXACT && hit.t < 1e9f && (hit.inst & kHitTied) != 0u) {
                TieLeaves tl;
                tie_leaves<COUNT, W4>(sc, sx, st, o3, d3, hit.t, tl, cnt);
                TraceHit xh;
                if (tl.n > 0 && tie_walk(sx, sh.instances, sx.tlas_current, tl, o3, d3, xh) && xh.t == hit.t) {
                    // (xh.t differs only if a float accident kept the reference's walk from the tying leaves: then the native
                    // walk's answer stands)
                    hit = xh;
                    hit.tri = sx.tri_native[xh.tri];   // shading and reach records are in the native order
                }
            }
            // (the ray-segment count the host reads is the sum of the queue sizes: this path's later segments are in no queue)
            if (bounce > first) atomicAdd(&wb.qcount[(size_t)bounce * kSegments], 1u);
            const float4 ha = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
            const uint32_t hb = (hit.inst & ~kHitTied) | (hit.front ? 0x80000000u : 0u);
            bool unreachable;
            float4 no, nd, nt;
            if (!shade_entry<COUNT>(sh, wb, dm, fp, cam_far, bounce, ro, rd, tin, ha, hb, false, unreachable, no, nd, nt, cnt)) break;
            ro = no;
            rd = nd;
            tin = nt;
        }
    }
    if (COUNT) flush_counters(cnt, counters);
}

// ---- the sky cells of culled pixels (wf2_accumulate; the argument is there) -------------------------------------------------
constexpr float kCellMargin = 0.01f;
// values * 255 + 0.5 of the sky along the ray through a raster position, per channel; false: the camera block does not behave
// there (the clip-space w changes sign against `w_ref`, or a NaN).  (raster_direction_approx: reciprocal estimates instead of six
// divisions and a square root -- a few ulp of d.y, i.e. 1e-5 of a cell against kCellMargin = 1e-2.)
__device__ __forceinline__ bool sky_cells_at(const RefCamera& cam, float two_over_w, float two_over_h, float fx, float fy, float w_ref, float& w_out,
                                             float v[3], f3& d)
{
    d = raster_direction_approx(cam, two_over_w, two_over_h, fx, fy, w_out);
    const f3 c = mk3(0.0f, 0.0f, 0.0f) + mk3(1.0f, 1.0f, 1.0f) * sample_sky(d);
    v[0] = clamp_(c.x, 0.0f, 1.0f) * 255.0f + 0.5f;
    v[1] = clamp_(c.y, 0.0f, 1.0f) * 255.0f + 0.5f;
    v[2] = clamp_(c.z, 0.0f, 1.0f) * 255.0f + 0.5f;
    return (w_out * w_ref > 0.0f) && (d.y == d.y);   // (callers that compare the signs themselves pass w_ref = 1)
}
// how far d.y can leave the range of the four corner values inside the quadrilateral, in cells; 1e30 when it is too wide to bound
__device__ __forceinline__ float sky_interior_excess(const f3 c[4])
{
    float chord2 = 0.0f, ay = 0.0f;
    for (int i = 0; i < 4; i++) {
        ay = fmax_(ay, __builtin_fabsf(c[i].y));
        for (int j = i + 1; j < 4; j++) {
            const f3 e = c[i] - c[j];
            chord2 = fmax_(chord2, e.x * e.x + e.y * e.y + e.z * e.z);
        }
    }
    if (!(chord2 < 0.2f)) return 1e30f;                        // (also a NaN)
    const float theta2 = 1.1f * chord2;                        // theta <= 0.5: theta^2 < 1.03 chord^2
    const float m = fmin_(1.0f, ay + __builtin_sqrtf(theta2));  // max |d.y| inside: a corner's, plus at most theta
    return 6.4f * 0.25f * theta2 * m;
}

// The TILE-level test of wf2_accumulate as a pass of its own, one LANE per 8 x 8 tile instead of one wave (round 5): do the four
// corner rays of the tile agree on one rgba8 sky cell per channel, with the margins of the argument in wf2_accumulate?  Then every
// culled pixel of the tile has that value for all its frames: tile_cell = 0x80000000 | r | g << 8 | b << 16; else 0 (the tile's
// culled pixels are decided one by one, or take the exact route).  Depends on the camera, the image size and the partition only:
// the host runs it when one of those changes (jpt_capi.cpp), not per render -- 32 400 lanes once per camera instead of 32 400 waves
// x ~300 instructions in every accumulation.
__global__ __launch_bounds__(kBlock) void wf2_sky_tiles(Wf2Dims dm, FrameParams fp, RefCamera cam, uint32_t* __restrict__ tile_cell)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= (uint32_t)dm.full_tiles_x * (uint32_t)dm.full_tiles_y) return;
    const uint32_t fty = fdiv(t, dm.by_full_tiles_x), ftx = t - fty * (uint32_t)dm.full_tiles_x;
    const int tx0 = (int)ftx * 8, ty0 = local_to_global_row((int)fty * 8, fp);
    const float two_over_w = 2.0f / (float)fp.width, two_over_h = 2.0f / (float)fp.height;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    f3 cd[4];
    float w0 = 1.0f;
    bool ok = true;
    for (int corner = 0; corner < 4; corner++) {
        float ww, v[3];
        (void)sky_cells_at(cam, two_over_w, two_over_h, (float)(tx0 + 8 * (corner & 1)), (float)(ty0 + 8 * (corner >> 1)), 1.0f, ww, v, cd[corner]);
        if (corner == 0) w0 = ww;
        ok = ok && (ww == ww) && (v[0] == v[0]) && (ww * w0 > 0.0f);   // no NaN, and the four w of one sign
        for (int k = 0; k < 3; k++) {
            lo[k] = fmin_(lo[k], v[k]);
            hi[k] = fmax_(hi[k], v[k]);
        }
    }
    const float excess = sky_interior_excess(cd);
    uint32_t cell = 0x80000000u;
    for (int k = 0; k < 3; k++) {
        const float l = lo[k] - excess, h = hi[k] + excess;
        const float c = __builtin_floorf(l);
        ok = ok && (l - c >= kCellMargin) && (h - c <= 1.0f - kCellMargin) && (h - l < 0.5f);
        cell |= ((uint32_t)c & 255u) << (8 * k);
    }
    tile_cell[t] = ok ? cell : 0u;
}

// ---- per pixel: frames in order -> accumulation buffer, display image, depth ----------------------------------

__global__ __launch_bounds__(kBlock) void wf2_accumulate(Wf2Buffers wb, Wf2Dims dm, FrameParams fp, RefCamera cam, SkyCull cull,
                                                         float4* __restrict__ accum, uint32_t* __restrict__ ldr,
                                                         float* __restrict__ depth_out, const uint32_t* __restrict__ tile_cell)
{
    // one thread per pixel of the context's share of the image, tile by tile (a wave = one 8 x 8 tile); `slot` is the
    // pixel's place in the window
    constexpr int kSharedFrames = 16;                       // most frames per render the shared exact route below holds
    __shared__ uint32_t s_who[kBlock / 64][64];            // per wave: the pixels that need it, px | py << 16
    __shared__ uint32_t s_val[kBlock / 64][64 * kSharedFrames];   // ... and their frames' rgba8 sky values
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t full_slot = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t ftile = full_slot >> 6, flane = full_slot & 63u;
    const uint32_t fty = fdiv(ftile, dm.by_full_tiles_x), ftx = ftile - fty * (uint32_t)dm.full_tiles_x;
    if ((int)fty >= dm.full_tiles_y) return;   // (the whole wave)
    const int px = (int)(ftx * 8u + (flane & 7u)), ly = (int)(fty * 8u + (flane >> 3));
    const bool valid = px < fp.width && ly < fp.local_rows;   // (lanes past the image's edge still help with the shared route)
    const int wtx = (int)ftx - dm.tile_x0, wty = (int)fty - dm.tile_y0;
    const bool in_window = wtx >= 0 && wtx < dm.tiles_x && wty >= 0 && wty < dm.tiles_y;
    const uint32_t in_tile = flane;
    const uint32_t slot = in_window ? ((uint32_t)wty * (uint32_t)dm.tiles_x + (uint32_t)wtx) * 64u + in_tile : 0u;
    const size_t idx = valid ? (size_t)ly * fp.width + px : 0;
    // fp.frame_count = ProgressiveRendering frame_count of the FIRST frame of this render
    f3 sum = mk3(0.0f, 0.0f, 0.0f);
    bool have_prev = fp.frame_count > 1;
    if (have_prev && valid) {
        const float4 prev = stream_ld4(&accum[idx]);
        sum = mk3(prev.x, prev.y, prev.z);
    }
    f3 last = mk3(0.0f, 0.0f, 0.0f);
    // the primary launch neither traced nor stored the paths of a sky-culled pixel: their radiance is the sky along the
    // primary ray of (x, y, frame) (main.glsl:380,395-397 with throughput 1), made up here
    const int py = local_to_global_row(ly, fp);
    const bool culled = valid && (!in_window || sky_culled(cull, px, py));
    // A culled pixel's frames all see the sky, whose colour depends on the ray's d.y alone and changes by a hundredth of an
    // rgba8 step across a pixel: in REF_LDR8 mode nearly every pixel has ONE rgba8 sky value for all its frames.  The
    // jittered sample of a frame lies on the quarter circle (px + cos, py + sin), inside the pixel's square.  A pinhole
    // maps raster lines to great circles, so the directions through the square fill the spherical quadrilateral of its four
    // corner rays, and along a great circle d.y = A cos(s + phi): f'' = -f.  Every point of the quadrilateral lies on an
    // arc between two points of its edges, so d.y leaves the range of the four corner values by at most
    //     2 * (theta^2 / 8) * max |d.y|        (theta: the quadrilateral's diameter; two levels of interpolation)
    // -- nothing for the narrow pixels of the benchmark cameras (1e-6), but NOT nothing for a wide lens looking at the
    // zenith, where d.y has its maximum INSIDE the square (0.06 of a cell for an 8 x 8 tile at fov 150).  The
    // corner range is therefore widened by that bound, in cell units (d value / d d.y <= 255 * 0.05 * 0.5 = 6.4,
    // main.glsl:189-192), before it is asked to lie inside ONE integer cell of value * 255 + 0.5, `kCellMargin` away from
    // the cell's ends (float rounding of the ray set-up: 1e-5 of a cell); a quadrilateral wider than half a radian is not
    // tried.  When that holds every frame quantises to that cell and the eight primary rays -- seed, sincos, 4 x 4
    // transform, three divisions, a normalisation each -- need not be made.  Pixels near a cell boundary (a few per cent:
    // horizontal bands) and cameras whose clip-space w changes sign inside the pixel take the exact per-frame route below.
    bool sky_constant = false;
    f3 sky_value = mk3(0.0f, 0.0f, 0.0f);
    uint32_t sky_word = 0u;   // ... as the rgba8 word it is converted from
    const float two_over_w = 2.0f / (float)fp.width, two_over_h = 2.0f / (float)fp.height;
    const bool want_cells = fp.accum_mode == 0 && fp.n_frames > 1;
    // First for the whole TILE at once (a wave is one 8 x 8 tile, eight consecutive image rows): do its four corner rays agree on the
    // cells?  Then every culled pixel of the tile has that value.  (The same argument over eight pixels instead of one; theta is
    // eight times larger, so wide lenses fail here and pass pixel by pixel.)  Decided by wf2_sky_tiles, one lane per tile, while
    // the render's paths were traced: here one word per wave.
    bool tile_constant = false;
    if (want_cells && tile_cell != nullptr && __any(culled)) {
        const uint32_t tc = tile_cell[__builtin_amdgcn_readfirstlane((int)ftile)];
        if (tc & 0x80000000u) {
            tile_constant = true;
            if (culled) {
                sky_constant = true;
                sky_word = tc & 0xffffffu;
                sky_value = mk3(from_unorm8(tc & 255u), from_unorm8((tc >> 8) & 255u), from_unorm8((tc >> 16) & 255u));
            }
        }
    }
    // ... then, in the tiles that straddle a cell boundary, pixel by pixel.  (Sharing the tile's 9 x 9 corners through LDS -- two
    // passes of the set-up instead of four -- left the kernel at 49.5 us and cost queued renders 1-5 %: profiles/r04/r04ak_acc_corners.txt)
    if (culled && want_cells && !tile_constant) {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        bool sane = true;
        float w_first = 1.0f;
        f3 cd[4];
        for (int corner = 0; corner < 4; corner++) {
            float ww, v[3];
            const bool ok = sky_cells_at(cam, two_over_w, two_over_h, (float)(px + (corner & 1)), (float)(py + (corner >> 1)), w_first, ww, v, cd[corner]);
            if (corner == 0) w_first = ww;
            sane = sane && (corner == 0 ? (ww == ww) : ok);
            for (int k = 0; k < 3; k++) {
                lo[k] = fmin_(lo[k], v[k]);
                hi[k] = fmax_(hi[k], v[k]);
            }
        }
        const float excess = sky_interior_excess(cd);
        bool same = sane;
        float cell[3];
        for (int k = 0; k < 3; k++) {
            lo[k] -= excess;
            hi[k] += excess;
            cell[k] = __builtin_floorf(lo[k]);
            same = same && (lo[k] - cell[k] >= kCellMargin) && (hi[k] - cell[k] <= 1.0f - kCellMargin) && (hi[k] - lo[k] < 0.5f);
        }
        if (same) {
            sky_constant = true;
            sky_word = ((uint32_t)cell[0] & 255u) | (((uint32_t)cell[1] & 255u) << 8) | (((uint32_t)cell[2] & 255u) << 16);
            sky_value = mk3(from_unorm8(sky_word & 255u), from_unorm8((sky_word >> 8) & 255u), from_unorm8((sky_word >> 16) & 255u));
        }
    }
    // The culled pixels that did not pass (near a cell boundary) need their frames' exact values -- eight primary rays
    // each.  A wave runs that code for all 64 lanes as soon as ONE pixel of its tile needs it, frame after frame; instead
    // the tile's (pixel, frame) pairs that need it are dealt to the wave's 64 lanes, one pair each, so a tile with eight
    // such pixels makes one pass (64 pairs), not eight.  The values go through LDS to the pixel that sums them, in frame
    // order.  (REF_LDR8, 2..kSharedFrames frames per render; otherwise every lane walks its own frames, as before.)
    const bool shared_route = fp.accum_mode == 0 && fp.n_frames > 1 && fp.n_frames <= kSharedFrames;
    const bool slow = culled && !sky_constant && shared_route;
    uint32_t my_rank = 0;
    {
        const unsigned long long sm = __ballot(slow);
        if (sm) {
            my_rank = lanes_below(sm, lane);
            if (slow) s_who[wave][my_rank] = (uint32_t)px | ((uint32_t)py << 16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t total = (uint32_t)__popcll(sm) * (uint32_t)fp.n_frames;
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t item = base + (uint32_t)lane;
                if (item < total) {
                    const uint32_t p = fdiv(item, dm.by_frames), f = item - p * (uint32_t)fp.n_frames;
                    const uint32_t who = s_who[wave][p];
                    uint32_t sx, sy;
                    const Ray ray = primary_ray(cam, fp.width, fp.height, (int)(who & 0xffffu), (int)(who >> 16), fp.frame_index + f, sx, sy);
                    const f3 c = mk3(0.0f, 0.0f, 0.0f) + mk3(1.0f, 1.0f, 1.0f) * sample_sky(ray.d);
                    s_val[wave][item] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16);   // rgba8 store of main.glsl:434
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!valid) return;
    // the frames of group g are a block of their own in rad / fin8: [slot][frame of the group] behind the earlier groups' blocks
    const int g_base = fp.n_frames / dm.acc_groups, g_extra = fp.n_frames % dm.acc_groups;
    int g = 0, g_f0 = 0, g_nf = g_base + (g_extra > 0 ? 1 : 0);
    // The usual case first -- rgba8 samples, one frame group, 4 / 8 / 12 / 16 frames: a pixel's frames are 16 to 64 consecutive bytes of
    // fin8, fetched as one to four 16-byte loads that are all in flight together.  (The loop below waits for a 4-byte load per frame,
    // eight trips to the cache one after another: the kernel was bound by that latency, not by its 53 MB -- round 5.)
    const bool packed_frames = fp.accum_mode == 0 && dm.acc_groups == 1 && fp.n_frames >= 4 && fp.n_frames <= kSharedFrames && (fp.n_frames & 3) == 0;
    if (packed_frames) {
        // every lane's frames as rgba8 words first -- the window's pixels from fin8, a culled pixel's from its one sky cell or from
        // the wave's shared exact values -- then ONE conversion-and-add per frame for all of them
        uint4 q[kSharedFrames / 4];
        const uint4* mine = reinterpret_cast<const uint4*>(wb.fin8 + (size_t)slot * (size_t)fp.n_frames);
        const uint4* shared = reinterpret_cast<const uint4*>(&s_val[wave][my_rank * (uint32_t)fp.n_frames]);
#pragma unroll
        for (int c = 0; c < kSharedFrames / 4; c++) {
            q[c] = make_uint4(sky_word, sky_word, sky_word, sky_word);
            if (c * 4 < fp.n_frames) {
                if (!culled) q[c] = stream_ldu4(&mine[c]);
                else if (slow) q[c] = shared[c];
            }
        }
#pragma unroll
        for (int c = 0; c < kSharedFrames / 4; c++) {
            if (c * 4 >= fp.n_frames) break;
            const uint32_t qs[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f3 cur = mk3(from_unorm8(qs[j] & 255u), from_unorm8((qs[j] >> 8) & 255u), from_unorm8((qs[j] >> 16) & 255u));
                last = cur;
                sum = have_prev ? cur + sum : cur;  // progressive_rendering.glsl:34-36
                have_prev = true;
            }
        }
    }
    for (int f = 0; f < fp.n_frames && !packed_frames; f++) {
        if (f >= g_f0 + g_nf) {
            g++;
            g_f0 += g_nf;
            g_nf = g_base + (g < g_extra ? 1 : 0);
        }
        const size_t at = (size_t)g_f0 * dm.slots_per_frame + (size_t)slot * (size_t)g_nf + (size_t)(f - g_f0);
        f3 cur;
        if (sky_constant) {
            cur = sky_value;
            last = cur;
        } else if (slow) {   // rgba8 load of progressive_rendering.glsl:33
            const uint32_t q = s_val[wave][my_rank * (uint32_t)fp.n_frames + (uint32_t)f];
            cur = mk3(from_unorm8(q & 255u), from_unorm8((q >> 8) & 255u), from_unorm8((q >> 16) & 255u));
            last = cur;
        } else if (culled) {
            uint32_t sx, sy;
            const Ray ray = primary_ray(cam, fp.width, fp.height, px, py, fp.frame_index + (uint32_t)f, sx, sy);
            cur = mk3(0.0f, 0.0f, 0.0f) + mk3(1.0f, 1.0f, 1.0f) * sample_sky(ray.d);
            if (fp.accum_mode == 0) {  // rgba8 store of main.glsl:434, load of progressive_rendering.glsl:33
                last = mk3(from_unorm8(unorm8(cur.x)), from_unorm8(unorm8(cur.y)), from_unorm8(unorm8(cur.z)));
                cur = last;
            } else {
                last = cur;
            }
        } else if (fp.accum_mode == 0) {
            const uint32_t q = stream_ldu(&wb.fin8[at]);
            cur = mk3(from_unorm8(q & 255u), from_unorm8((q >> 8) & 255u), from_unorm8((q >> 16) & 255u));
            last = cur;   // (display_mode 1 shows the last frame's rgba8 image itself: quantising it again gives the same bytes)
        } else {
            const float4 r = stream_ld4(&wb.rad[at]);
            cur = mk3(r.x, r.y, r.z);
            last = cur;
        }
        sum = have_prev ? cur + sum : cur;  // progressive_rendering.glsl:34-36
        have_prev = true;
    }
    if (fp.n_frames > 0) {
        stream_st4(&accum[idx], make_float4(sum.x, sum.y, sum.z, 1.0f));
        const float fc = (float)(fp.frame_count + (uint32_t)fp.n_frames - 1u);
        const f3 col = fp.display_mode == 1 ? last : aces_film(mk3(sum.x / fc, sum.y / fc, sum.z / fc) * 1.0f);
        stream_stu(&ldr[idx], unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u);
        if (depth_out) {
            const float dist = culled ? cam.far_ : wb.first_depth[slot];
            stream_stf(&depth_out[idx], cam.far_ / (cam.far_ - cam.near_) * (1.0f - cam.near_ / dist));  // main.glsl:432
        }
    }
}

// the window of a render (local tiles): x0, y0, nx, ny
struct TileWindow {
    int x0, y0, nx, ny;
};
TileWindow full_window(int width, int local_rows) { return TileWindow{0, 0, (width + 7) / 8, (local_rows + 7) / 8}; }

// The tile-aligned bounding rectangle of the cull's screen rectangles, in this context's local tile rows (8-row strips
// s with s % world == rank: local tile row t is image tile row t * world + rank).  Outside it every pixel is
// sky-culled, so no path starts there.  Nothing visible at all: one tile (its pixels are culled one by one).
TileWindow cull_window(const SkyCull& cull, const FrameParams& fp)
{
    const TileWindow full = full_window(fp.width, fp.local_rows);
    if (cull.n < 0) return full;
    const TileWindow none{0, 0, 1, 1};
    int gx0 = INT32_MAX, gy0 = INT32_MAX, gx1 = INT32_MIN, gy1 = INT32_MIN;
    for (int k = 0; k < 4 && k < cull.n; k++) {
        if (cull.x0[k] > cull.x1[k] || cull.y0[k] > cull.y1[k]) continue;
        gx0 = std::min(gx0, cull.x0[k]);
        gy0 = std::min(gy0, cull.y0[k]);
        gx1 = std::max(gx1, cull.x1[k]);
        gy1 = std::max(gy1, cull.y1[k]);
    }
    gx0 = std::max(gx0, 0);
    gy0 = std::max(gy0, 0);
    gx1 = std::min(gx1, fp.width - 1);
    gy1 = std::min(gy1, fp.height - 1);
    if (gx0 > gx1 || gy0 > gy1) return none;
    const int world = fp.world > 0 ? fp.world : 1;
    const int ty0 = gy0 / 8 - fp.rank, ty1 = gy1 / 8 - fp.rank;   // image tile rows, shifted so that this rank's are the multiples of world
    if (ty1 < 0) return none;
    const int lt0 = ty0 <= 0 ? 0 : (ty0 + world - 1) / world, lt1 = std::min(ty1 / world, full.ny - 1);
    if (lt0 > lt1) return none;
    return TileWindow{gx0 / 8, lt0, gx1 / 8 - gx0 / 8 + 1, lt1 - lt0 + 1};
}

Wf2Dims make_dims(int width, int local_rows, int n_frames, const TileWindow& win)
{
    Wf2Dims dm;
    dm.tile_x0 = win.x0;
    dm.tile_y0 = win.y0;
    dm.tiles_x = win.nx;
    dm.tiles_y = win.ny;
    dm.full_tiles_x = (width + 7) / 8;
    dm.full_tiles_y = (local_rows + 7) / 8;
    dm.by_full_tiles_x = make_fastdiv((uint32_t)dm.full_tiles_x);
    dm.tiles_per_frame = (uint32_t)dm.tiles_x * (uint32_t)dm.tiles_y;
    dm.slots_per_frame = dm.tiles_per_frame * 64u;
    dm.n_chunks = dm.tiles_per_frame * (uint32_t)n_frames;
    // Chunks are dealt to the segments in runs of 2^run_shift consecutive chunks -- consecutive parts of one tile, whose rays stay
    // alike for longer -- but runs also unbalance the segments: single chunks up to a few hundred per segment, runs of two / four
    // only for the largest renders (3840 x 2160 x 16: 1 157 chunks per segment).  Swept in rounds 1-4: LAB_NOTEBOOK.md.
    const uint32_t per_segment = dm.n_chunks / kSegments;
    const uint32_t run_shift = per_segment >= 512u ? 2u : (per_segment >= 256u ? 1u : 0u);
    dm.run_shift = run_shift;
    const uint32_t n_runs = (dm.n_chunks + (1u << run_shift) - 1u) >> run_shift;
    dm.seg_cap = (((n_runs + kSegments - 1u) / kSegments) << run_shift) * 64u;
    dm.by_tiles_x = make_fastdiv((uint32_t)dm.tiles_x);
    dm.by_frames = make_fastdiv((uint32_t)(n_frames > 0 ? n_frames : 1));
    return dm;
}

}  // namespace

size_t wf2_sky_tile_count(int width, int local_rows) { return (size_t)((width + 7) / 8) * (size_t)((local_rows + 7) / 8); }
void launch_sky_tiles(hipStream_t stream, const FrameParams& fp, const RefCamera& cam, uint32_t* tile_cell)
{
    const Wf2Dims dm = make_dims(fp.width, fp.local_rows, 1, full_window(fp.width, fp.local_rows));
    const uint32_t n_tiles = (uint32_t)dm.full_tiles_x * (uint32_t)dm.full_tiles_y;
    if (n_tiles) hipLaunchKernelGGL(wf2_sky_tiles, dim3((n_tiles + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, dm, fp, cam, tile_cell);
}
uint32_t wf2_segments() { return kSegments; }
uint32_t trace_stack_capacity() { return (uint32_t)(kStackLds + kStackSpill); }

// Frame groups.  Every launch of the pipeline ends with a tail: a few long rays in a few waves while the rest of the
// chip has nothing left to do (a ray's latency under full load is ~20 us on average, ~100 us for the longest; C3's ten
// launch boundaries cost 0.5 ms of 1.85 -- render time is 0.48 + 0.17 * spp ms).  Balancing the queues does not
// help (tried: equal ray counts per block, same time) and neither does fusing the bounce loop into one persistent
// kernel (the barriers move into the blocks).  What helps is having other work ready when a launch drains: the
// frames of a render are split into groups, each group runs the pipeline on its own stream with its own queues,
// and the hardware fills the slots one group's kernel frees with the blocks of the other group's next kernel.
// Paths never cross groups and the per-pixel accumulation still reads the frames in order: results are unchanged.
// Measured on one MI355X (ms per blocking render, 1 / 2 / 3 / 4 groups of full-width launches): 3840x2160x16 spp 13.2 /
// 11.9 / 12.2 / 13.1; 1920x1080x32 spp 5.92 / 5.38 / 5.75 / 6.18; x8 spp (C3) 1.81 / 1.80 / 2.09 / 2.54; x4 spp 1.17 /
// 1.31: every extra launch costs ~25 us.  Two groups whose tracing launches are HALF as wide (two chained segments per
// block, wf2_trace) do better, because the two groups' launches then really run side by side: C3 1.76 -> 1.61,
// close-up 6.89 -> 6.22, 16 spp 3.11 -> 2.82, instanced scene 4.56 -> 4.03; 1280x720x4 spp 0.69 -> 0.71 (not used
// below 12 M paths).
constexpr int kMaxGroups = 4;
static int frame_groups(int n_frames, bool serial, size_t paths)
{
    const int forced = [] {  // JPT_GROUPS=n overrides the rule (tuning runs, tests)
        const int g = tuning().groups;
        return g < 0 ? 0 : (g > kMaxGroups ? kMaxGroups : g);
    }();
    if (serial || n_frames < 2) return 1;
    const int wanted = forced ? forced : (paths >= ((size_t)12 << 20) ? 2 : 1);
    return n_frames < wanted ? n_frames : wanted;
}
static void group_frames(int n_frames, int groups, int g, int& first, int& count)
{
    const int base = n_frames / groups, extra = n_frames % groups;
    first = g * base + (g < extra ? g : extra);
    count = base + (g < extra ? 1 : 0);
}

// Set-aside records per group: a path is set aside at most once, and few are (float cracks of the reference's boxes: a
// handful per render; exact distance ties: rare outside scenes built of coincident geometry): 1/64 of the paths, at least
// 65 536 (2 MB) and never more than the paths, instead of one 32-byte record per path (0.5 GB per pipeline slot at C3).
// Overflow is counted, not silent: jpt_stats.set_aside_dropped.
static uint32_t redo_capacity(size_t paths)
{
    if (tuning().set_aside_cap >= 0) return (uint32_t)std::min<size_t>((size_t)tuning().set_aside_cap, paths);   // (tests)
    const size_t c = std::min(paths, std::max<size_t>(paths / 64u, 65536u));
    return (uint32_t)(c > 0x7fffffffu ? 0x7fffffffu : c);
}

size_t wf2_workspace_bytes(int width, int local_rows, int n_frames, int max_bounces)
{
    // sized for every layout a render of this size may use (the group count depends on kernel timing, on the
    // previous render's ray count and on JPT_GROUPS): the largest of 1..kMaxGroups groups
    size_t worst = 0;
    for (int groups = 1; groups <= kMaxGroups && groups <= (n_frames < 1 ? 1 : n_frames); groups++) {
        size_t b = 0;
        for (int g = 0; g < groups; g++) {
            int f0, nf;
            group_frames(n_frames, groups, g, f0, nf);
            const Wf2Dims dm = make_dims(width, local_rows, nf, full_window(width, local_rows));
            const size_t q = (size_t)dm.seg_cap * kSegments;                  // queue entries
            const size_t paths = (size_t)dm.slots_per_frame * (size_t)nf;
            b += q * sizeof(float4) * 6 + 6 * 256;    // two ray queues (o, d, throughput)
            b += q * sizeof(float4) + 256;            // hit_a
            b += q * sizeof(uint32_t) + 256;          // hit_b
            b += paths * sizeof(float4) + 256;        // thr
            b += ((size_t)(max_bounces + 2) * kSegments + 64) * sizeof(uint32_t) + 256;   // queue sizes + the set-aside counts
            b += (size_t)redo_capacity(paths) * 2 * sizeof(float4) + 256;    // set-aside records
        }
        const Wf2Dims all = make_dims(width, local_rows, n_frames, full_window(width, local_rows));
        b += (size_t)all.slots_per_frame * (size_t)n_frames * sizeof(float4) + 256;  // rad: a block per group, each [slot][frame of the group]
        b += (size_t)all.slots_per_frame * (size_t)n_frames * sizeof(uint32_t) + 256;  // fin8
        b += (size_t)all.slots_per_frame * sizeof(float) + 256;
        worst = b > worst ? b : worst;
    }
    return worst + 17 * 256;
}

namespace {
__global__ void add_queue_counts(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
}  // namespace

void launch_wf2_render(hipStream_t stream, const DeviceScene& ds, const FrameParams& fp, const RefCamera& cam, void* workspace,
                       float4* accum, uint32_t* ldr, float* depth, DevCounters* counters, hipEvent_t* trace_events,
                       const Wf2Async& async)
{
    const TileWindow window = cull_window(async.cull, fp);
    const Wf2Dims dm_all = make_dims(fp.width, fp.local_rows, fp.n_frames, window);
    if (dm_all.n_chunks == 0) return;
    char* w = reinterpret_cast<char*>(workspace);
    auto carve = [&](size_t bytes) {
        void* p = w;
        w += (bytes + 255) & ~(size_t)255;
        return p;
    };
    const int nq = fp.max_bounces + 2;
    // per-launch events and event counters want the launches one after another
    const int groups = frame_groups(fp.n_frames, trace_events != nullptr || counters != nullptr || !async.aux_stream[0],
                                    (size_t)fp.width * (size_t)fp.local_rows * (size_t)fp.n_frames);
    Wf2Buffers gb[kMaxGroups];
    Wf2Dims gdm[kMaxGroups];
    FrameParams gfp[kMaxGroups];
    for (int g = 0; g < groups; g++) {  // group 0 first: the host reads it
        gb[g].qcount = (uint32_t*)carve(((size_t)nq * kSegments + 64) * sizeof(uint32_t));
        gb[g].redo_count = gb[g].qcount + (size_t)nq * kSegments;
    }
    float4* rad_all = (float4*)carve((size_t)dm_all.slots_per_frame * (size_t)fp.n_frames * sizeof(float4));
    uint32_t* fin8_all = (uint32_t*)carve((size_t)dm_all.slots_per_frame * (size_t)fp.n_frames * sizeof(uint32_t));
    float* first_depth = (float*)carve((size_t)dm_all.slots_per_frame * sizeof(float));
    for (int g = 0; g < groups; g++) {
        int f0, nf;
        group_frames(fp.n_frames, groups, g, f0, nf);
        gdm[g] = make_dims(fp.width, fp.local_rows, nf, window);
        const size_t q = (size_t)gdm[g].seg_cap * kSegments;
        const size_t paths = (size_t)gdm[g].slots_per_frame * (size_t)nf;
        Wf2Buffers& wb = gb[g];
        wb.ray_o[0] = (float4*)carve(q * sizeof(float4));
        wb.ray_o[1] = (float4*)carve(q * sizeof(float4));
        wb.ray_d[0] = (float4*)carve(q * sizeof(float4));
        wb.ray_d[1] = (float4*)carve(q * sizeof(float4));
        wb.thr_q[0] = (float4*)carve(q * sizeof(float4));
        wb.thr_q[1] = (float4*)carve(q * sizeof(float4));
        wb.hit_a = (float4*)carve(q * sizeof(float4));
        wb.hit_b = (uint32_t*)carve(q * sizeof(uint32_t));
        wb.redo_cap = redo_capacity(paths);
        wb.redo_rec = (float4*)carve((size_t)wb.redo_cap * 2 * sizeof(float4));
        wb.thr = (float4*)carve(paths * sizeof(float4));
        wb.rad = rad_all + (size_t)f0 * dm_all.slots_per_frame;  // this group's block ([slot][frame of the group]: path ids) behind the earlier groups'
        wb.fin8 = fin8_all + (size_t)f0 * dm_all.slots_per_frame;
        wb.first_depth = first_depth;
        gfp[g] = fp;
        gfp[g].frame_index = fp.frame_index + (uint32_t)f0;
        gfp[g].n_frames = nf;
        gfp[g].depth_frame = (fp.depth_frame >= 0 && f0 + nf == fp.n_frames) ? nf - 1 : -1;  // the render's last frame writes the depth image (when there is one: jpt_set_outputs)
    }

    const bool w4 = ds.use4;
    WideSceneDev sc;
    sc.blas_nodes = ds.blas_nodes;
    sc.tlas_nodes = ds.tlas_nodes;
    sc.nodesq = ds.nodesq;
    sc.tris = ds.wide_tris;
    sc.instances = w4 ? ds.wide_instances4 : ds.wide_instances;
    sc.tlas_root = w4 ? ds.tlas_root4 : ds.tlas_root;
    sc.n_instances = ds.n_instances;
    sc.reach_tri = ds.reach_tri;
    sc.reach_inst = ds.reach_inst;
    SceneShading sh = ds.shading();
    const dim3 block(kBlock);
    const WfTune tune{Tuning::refill_idle, Tuning::refill_idle, Tuning::node_min_lanes, Tuning::leaf_min_lanes, Tuning::inst_min_lanes, Tuning::phase_frac16, tuning().tail_rounds, tuning().tail_lanes};

    const int chain = [&] {  // two frame groups share the chip: half-width launches
        const int c = groups == 2 ? 2 : async.trace_chain;
        return c < 1 ? 1 : (c > kMaxChain ? kMaxChain : c);
    }();
    const dim3 tgrid((kSegments + (uint32_t)chain - 1u) / (uint32_t)chain);
    const dim3 pgrid(kSegments);
    // The waves of a tracing launch finish their last, very long walks themselves, all lanes on one ray (the TAIL instantiations,
    // coop_walk_call): scenes large enough to have such walks (>= 200 000 triangles); JPT_TAIL=0 never, 1 on every scene.
    const bool tail = w4 && (tuning().tail < 0 ? ds.n_tris >= 200000u : tuning().tail != 0);
    // the pipeline of one group on one stream
    auto run_group = [&](hipStream_t st, const Wf2Buffers& wb, const Wf2Dims& dm, const FrameParams& gp, hipEvent_t* ev) {
        // (the queue sizes of bounces >= 1 and the set-aside counts start from zero: wf2_primary clears them)
        // Blocks go to the 8 XCDs round-robin by linear index (y * grid.x + x), and the chunks of a segment are far from
        // alike (the first ones are full, the last ones empty): with grid.x a multiple of 8 every XCD would always get
        // the same chunk position.  An odd grid.x deals every position to every XCD (capping C3's grid.x from 37 to 8
        // cost 9 %).
        const dim3 sgrid(((dm.seg_cap + kBlock - 1) / kBlock) | 1u, kSegments);
        if (ev) (void)hipEventRecord(ev[0], st);
        if (tail) {
            if (counters) hipLaunchKernelGGL((wf2_primary<true, true, true>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
            else hipLaunchKernelGGL((wf2_primary<false, true, true>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
        } else if (counters) {
            if (w4) hipLaunchKernelGGL((wf2_primary<true, true>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
            else hipLaunchKernelGGL((wf2_primary<true, false>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
        } else {
            if (w4) hipLaunchKernelGGL((wf2_primary<false, true>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
            else hipLaunchKernelGGL((wf2_primary<false, false>), pgrid, block, 0, st, sc, wb, dm, gp, cam, tune, async.cull, counters);
        }
        if (ev) (void)hipEventRecord(ev[1], st);
        for (int b = 0; b <= gp.max_bounces; b++) {
            {
                // instantiations: the paths' last vertices without the BRDF code, scenes without a texture array without
                // the sampler code
                const bool last = b == gp.max_bounces;
                const bool tex = sh.tex != nullptr && sh.n_layers > 0 && sh.tex_res > 0;
                // 0 no texture array, 1 nearest filter, 2 linear filter (jpt.h: bit 1 of the sampler mode)
                const int texmode = !tex ? 0 : ((sh.sampler_mode & 2) ? 2 : 1);
#define JPT_LAUNCH_SHADE(C, L, T) hipLaunchKernelGGL((wf2_shade<C, L, T>), sgrid, block, 0, st, sh, wb, dm, gp, cam.far_, b, counters)
                if (counters) {
                    if (last) { if (texmode == 0) JPT_LAUNCH_SHADE(true, true, 0); else if (texmode == 1) JPT_LAUNCH_SHADE(true, true, 1); else JPT_LAUNCH_SHADE(true, true, 2); }
                    else      { if (texmode == 0) JPT_LAUNCH_SHADE(true, false, 0); else if (texmode == 1) JPT_LAUNCH_SHADE(true, false, 1); else JPT_LAUNCH_SHADE(true, false, 2); }
                } else {
                    if (last) { if (texmode == 0) JPT_LAUNCH_SHADE(false, true, 0); else if (texmode == 1) JPT_LAUNCH_SHADE(false, true, 1); else JPT_LAUNCH_SHADE(false, true, 2); }
                    else      { if (texmode == 0) JPT_LAUNCH_SHADE(false, false, 0); else if (texmode == 1) JPT_LAUNCH_SHADE(false, false, 1); else JPT_LAUNCH_SHADE(false, false, 2); }
                }
#undef JPT_LAUNCH_SHADE
            }
            if (b == gp.max_bounces) break;
            if (ev) (void)hipEventRecord(ev[2 * (b + 1)], st);
            if (tail) {
                if (counters) hipLaunchKernelGGL((wf2_trace<true, true, true>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
                else hipLaunchKernelGGL((wf2_trace<false, true, true>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
            } else if (counters) {
                if (w4) hipLaunchKernelGGL((wf2_trace<true, true>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
                else hipLaunchKernelGGL((wf2_trace<true, false>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
            } else {
                if (w4) hipLaunchKernelGGL((wf2_trace<false, true>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
                else hipLaunchKernelGGL((wf2_trace<false, false>), tgrid, block, 0, st, sc, wb, dm, b + 1, tune, chain, counters);
            }
            if (ev) (void)hipEventRecord(ev[2 * (b + 1) + 1], st);
        }
        if (sh.reach_tri) {  // the paths set aside because their hit is undecidable on the native tree: finished exactly
            const dim3 rgrid(256), rblock(64);   // (blocks past the set-aside count exit at once; more records than threads: grid-stride)
            const TieShadowDev& sx = ds.x;
            if (sx.ok && w4) {
                if (counters) hipLaunchKernelGGL((wf2_finish<true, true, true>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
                else hipLaunchKernelGGL((wf2_finish<false, true, true>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
            } else if (counters) {
                if (w4) hipLaunchKernelGGL((wf2_finish<true, true, false>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
                else hipLaunchKernelGGL((wf2_finish<true, false, false>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
            } else {
                if (w4) hipLaunchKernelGGL((wf2_finish<false, true, false>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
                else hipLaunchKernelGGL((wf2_finish<false, false, false>), rgrid, rblock, 0, st, sc, sx, sh, wb, dm, gp, cam.far_, counters);
            }
        }
    };

    if (groups == 1) {
        run_group(stream, gb[0], gdm[0], gfp[0], trace_events);
    } else {
        // fork: the helper streams start after everything already queued on the context's stream
        (void)hipEventRecord(async.fork, stream);
        for (int g = 1; g < groups; g++) (void)hipStreamWaitEvent(async.aux_stream[g - 1], async.fork, 0);
        // issue the groups' launches interleaved, so none of the streams runs ahead of the others on the host side
        // (run_group enqueues a whole pipeline; the hardware queues of the streams drain concurrently)
        for (int g = 0; g < groups; g++) run_group(g == 0 ? stream : async.aux_stream[g - 1], gb[g], gdm[g], gfp[g], nullptr);
        // join, then fold the other groups' queue sizes into group 0's (the host reads those for the ray count)
        for (int g = 1; g < groups; g++) {
            (void)hipEventRecord(async.join[g - 1], async.aux_stream[g - 1]);
            (void)hipStreamWaitEvent(stream, async.join[g - 1], 0);
        }
        const uint32_t nqc = (uint32_t)nq * kSegments + 2u;   // (and the two set-aside counts behind the queue sizes)
        for (int g = 1; g < groups; g++)
            hipLaunchKernelGGL(add_queue_counts, dim3((nqc + 255) / 256), dim3(256), 0, stream, gb[0].qcount, gb[g].qcount, nqc);
    }
    Wf2Buffers wb_all = gb[0];
    Wf2Dims dm_acc = dm_all;
    dm_acc.acc_groups = groups;
    wb_all.rad = rad_all;
    wb_all.fin8 = fin8_all;
    wb_all.first_depth = first_depth;
    // the accumulation touches the framebuffers: it waits for whatever ordered the context's renders before this one (before_acc)
    hipStream_t acc_stream = stream;
    if (async.before_acc) (void)hipStreamWaitEvent(acc_stream, async.before_acc, 0);
    const uint32_t ablocks = ((uint32_t)dm_all.full_tiles_x * (uint32_t)dm_all.full_tiles_y * 64u + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(wf2_accumulate, dim3(ablocks), block, 0, acc_stream, wb_all, dm_acc, fp, cam, async.cull, accum, ldr, depth, async.sky_tiles);
}

uint64_t wf2_pixels_outside_window(const SkyCull& cull, const FrameParams& fp)
{
    const TileWindow w = cull_window(cull, fp);
    const int64_t x0 = (int64_t)w.x0 * 8, x1 = std::min<int64_t>((int64_t)(w.x0 + w.nx) * 8, fp.width);
    const int64_t y0 = (int64_t)w.y0 * 8, y1 = std::min<int64_t>((int64_t)(w.y0 + w.ny) * 8, fp.local_rows);
    const int64_t inside = std::max<int64_t>(x1 - x0, 0) * std::max<int64_t>(y1 - y0, 0);
    return (uint64_t)((int64_t)fp.width * (int64_t)fp.local_rows - inside);
}

// the number of frame groups a blocking render of this size wants (helper streams permitting)
int wf2_wanted_groups(int n_frames, size_t paths) { return frame_groups(n_frames, false, paths); }

}  // namespace jpt
