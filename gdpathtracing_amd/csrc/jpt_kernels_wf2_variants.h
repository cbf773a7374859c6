// jpt_kernels_wf2_variants.h -- tracing launches that were built, measured and NOT kept as defaults; each stays behind its switch
// and under test as the measured form of its idea.  Included by jpt_kernels_wf2.hip, inside its namespace.
//
//   wf2_trace_rg   JPT_TRACE_REGROUP=1   round 4's regrouped bounce launch: ray state in LDS, compacted lists per kind of step.
//                                        Lanes 0.50 -> 0.78, twice the time (profiles/r04/r04c_regroup_ab.txt); its successor with
//                                        the two causes answered is wf2_trace_pool (jpt_kernels_wf2_pool.h, JPT_TRACE_REGROUP=2)
#pragma once

// ---- bounces >= 1, REGROUPED (JPT_TRACE_REGROUP=1; VERDICT r03 task 1) -------------------------------------------------
//
// wf2_trace binds a ray to a lane for its whole walk: a record step runs with the 28-39 of 64 lanes whose ray happens to want
// one, a leaf turn with 16-29.  Here a ray is bound to nothing.  A wave (= a block) keeps kRgPool rays'
// walk state in LDS -- level ray, closest hit, current record, a short stack: 56 + 4 * kRgStack bytes per ray -- and four byte
// lists of pool slots: rays that want a record step (NODE), a triangle leaf (LEAF), the world ray (TOP: an instance entry, or
// a TLAS record after an instance was left) and slots without a ray (FREE).  Each turn the wave takes up to 64 slots off ONE
// list, pulls their state into registers, runs that one kind of step with every lane occupied, writes the state back and
// files each slot under what its ray wants next (ballot + prefix popcount per list; everything is wave-synchronous: no
// barrier, no atomic but the queue cursor).  Per ray the sequence of steps -- the functions of jpt_trace_core.h, the same
// stack discipline -- is exactly wf2_trace's, so the hits are bit-identical; only which rays share a wave-level step changes.
// A finished ray's hit stays in its slot until the slot is refilled (written 64 at a time, like wf2_trace's late store).
#ifndef JPT_RG_POOL
#define JPT_RG_POOL 160
#endif
#ifndef JPT_RG_STACK
#define JPT_RG_STACK 8
#endif
constexpr int kRgPool = JPT_RG_POOL;      // rays in flight per wave (slot ids are bytes)
constexpr int kRgStack = JPT_RG_STACK;    // stack entries per ray in LDS ...
constexpr int kRgSpill = 64 - JPT_RG_STACK;   // ... and past them in global memory: 64 in all (the reference's own stack size, main.glsl:272,307)
static_assert(kRgPool >= 64 && kRgPool <= 256 && kRgPool % 4 == 0, "pool slots are addressed by bytes; a refill takes 64");
constexpr uint32_t kRgPending = 1u << 25;  // meta word: the slot holds a finished ray's hit that is not written yet
enum { kRgNode = 0, kRgLeaf = 1, kRgTop = 2, kRgFree = 3 };

struct RgPoolLds {
    float4 a[kRgPool];                 // level ray origin.xyz, hit.t
    float4 b[kRgPool];                 // level ray direction.xyz, current record (bits)
    float4 c[kRgPool];                 // hit.u, hit.v, hit.tri (bits), hit.inst | front << 31 (bits)
    uint32_t m[kRgPool];               // sp (0..7) | in_blas (8) | cur_inst (9..23) | kRgPending
    uint32_t loc[kRgPool];             // the ray's queue entry (world ray in, hit out)
    int32_t stack[kRgStack * kRgPool]; // [entry][slot]
    uint8_t list[4][256];              // rings of slot ids
};

constexpr uint32_t kRgMaxBlocks = 2u * kSegments;   // grid cap of a regrouped launch (the spill area is sized for it)
size_t wf2_rg_spill_bytes(uint32_t blocks) { return (size_t)blocks * kRgPool * kRgSpill * sizeof(int32_t); }

template <bool COUNT>
__global__ __launch_bounds__(64) void wf2_trace_rg(WideSceneDev sc, Wf2Buffers wb, Wf2Dims dm, int bounce, int chain, int waves_per_queue,
                                                   DevCounters* __restrict__ counters)
{
    __shared__ RgPoolLds P;
    using Walk = Traversal<COUNT, true>;
    const int lane = threadIdx.x;
    const uint32_t seg0 = (blockIdx.x / (uint32_t)waves_per_queue) * (uint32_t)chain;
    uint32_t end[kMaxChain];
    uint32_t n = 0;
    for (int k = 0; k < kMaxChain; k++) {
        if (k < chain && seg0 + (uint32_t)k < kSegments) n += wb.qcount[(size_t)bounce * kSegments + seg0 + (uint32_t)k];
        end[k] = n;
    }
    if (n == 0) return;
    uint32_t* __restrict__ cursor = wb.rg_cursor + (size_t)bounce * kSegments + seg0;
    const float4* __restrict__ qo = wb.ray_o[bounce & 1];
    const float4* __restrict__ qd = wb.ray_d[bounce & 1];
    int32_t* __restrict__ spill_base = wb.rg_spill + (size_t)blockIdx.x * kRgPool * kRgSpill;
    DevCounters cnt = {};
    // all slots free, nothing pending
    for (int s = lane; s < kRgPool; s += 64) {
        P.m[s] = 0u;
        P.list[kRgFree][s] = (uint8_t)s;
    }
    uint32_t hn = 0, hl = 0, ht = 0, hf = 0;            // list heads (free-running; the rings hold 256)
    uint32_t cn = 0, cl = 0, ct = 0, cf = kRgPool;      // list sizes
    // 64 queue entries are reserved one refill ahead: the atomic's round trip is over when its result is needed
    uint32_t next_start = 0;
    if (lane == 0) next_start = atomicAdd(cursor, 64u);
    bool exhausted = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    auto append = [&](int kind, bool pred, uint32_t id, uint32_t head, uint32_t& count) {
        const unsigned long long mk = __ballot(pred);
        if (mk) {
            if (pred) P.list[kind][(head + count + lanes_below(mk, lane)) & 255u] = (uint8_t)id;
            count += (uint32_t)__popcll(mk);
        }
    };
    auto stack_of = [&](uint32_t id) {
        return typename Walk::Stack{&P.stack[id], spill_base + (size_t)id * kRgSpill, kRgPool, kRgStack, kRgSpill};
    };
    auto write_hit = [&](uint32_t id) {
        const float4 a = P.a[id], c = P.c[id];
        const size_t loc = P.loc[id];
        wb.hit_a[loc] = make_float4(a.w, c.x, c.y, c.z);
        wb.hit_b[loc] = __float_as_uint(c.w);
    };
    // after a step: the next record off the ray's stack if the step left none, then the slot goes to the list of what the ray
    // wants next (pop_next of jpt_trace_core.h; a ray that left an instance wants the world ray back: TOP)
    auto file_slot = [&](bool on, uint32_t id, Walk& tr, const typename Walk::Stack& st) {
        bool fin = false, left_instance = false;
        if (on && !tr.have) {
            if (tr.sp == 0) {
                fin = true;
            } else {
                tr.cur = tr.pop(st);
                if (tr.cur == kSentinel) {
                    tr.in_blas = false;
                    left_instance = true;
                    if (tr.sp == 0) fin = true;
                    else tr.cur = tr.pop(st);
                }
            }
        }
        if (on) {
            P.b[id].w = __int_as_float(tr.cur);
            P.m[id] = (uint32_t)tr.sp | (tr.in_blas ? 256u : 0u) | (tr.cur_inst << 9) | (fin ? kRgPending : 0u);
        }
        const bool go = on && !fin;
        const bool to_top = go && !tr.in_blas && (tr.cur < 0 || left_instance);
        const bool to_node = go && tr.cur >= 0 && !to_top;
        const bool to_leaf = go && tr.cur < 0 && tr.in_blas;
        append(kRgNode, to_node, id, hn, cn);
        append(kRgLeaf, to_leaf, id, hl, cl);
        append(kRgTop, to_top, id, ht, ct);
        append(kRgFree, on && fin, id, hf, cf);
    };

    for (;;) {
        int kind;
        uint32_t take;
        if (!exhausted && cf >= 64u) {
            kind = kRgFree;
            take = 64u;
        } else {
            // a full chunk if any list has one, else the longest list
            const uint32_t best = cn >= cl ? (cn >= ct ? cn : ct) : (cl >= ct ? cl : ct);
            if (best == 0u) break;   // (nothing in flight, and no refill: the queue is exhausted)
            kind = cn >= 64u ? kRgNode : (cl >= 64u ? kRgLeaf : (ct >= 64u ? kRgTop : (best == cn ? kRgNode : (best == cl ? kRgLeaf : kRgTop))));
            const uint32_t have_n = kind == kRgNode ? cn : (kind == kRgLeaf ? cl : ct);
            take = have_n < 64u ? have_n : 64u;
        }
        const bool on = (uint32_t)lane < take;
        if (kind == kRgFree) {
            // refill: the reserved queue entries go into free slots (whose previous rays' hits are written out first)
            const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_start);
            const uint32_t avail = start < n ? (n - start < 64u ? n - start : 64u) : 0u;
            if (start + 64u >= n) exhausted = true;
            else if (lane == 0) next_start = atomicAdd(cursor, 64u);
            const bool mine = (uint32_t)lane < avail;
            uint32_t id = 0;
            bool trivial = false;
            if (mine) {
                id = P.list[kRgFree][(hf + (uint32_t)lane) & 255u];
                if (P.m[id] & kRgPending) write_hit(id);
                const uint32_t idx = start + (uint32_t)lane;
                uint32_t k = 0, first = 0;
                for (int j = 0; j < kMaxChain - 1; j++)
                    if (idx >= end[j]) {
                        k = (uint32_t)j + 1u;
                        first = end[j];
                    }
                const size_t loc = (size_t)(seg0 + k) * dm.seg_cap + (idx - first);
                const float4 ro = qo[loc], rd = qd[loc];
                // Traversal::begin: the walk starts at the TLAS root with the world ray; hit.t = 1e9 (main.glsl:354)
                trivial = sc.n_instances == 0u;
                P.a[id] = make_float4(ro.x, ro.y, ro.z, 1e9f);
                P.b[id] = make_float4(rd.x, rd.y, rd.z, __int_as_float(sc.tlas_root));
                P.c[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                P.m[id] = trivial ? kRgPending : 0u;
                P.loc[id] = (uint32_t)loc;
            }
            hf += avail;
            cf -= avail;
            append(kRgNode, mine && !trivial && sc.tlas_root >= 0, id, hn, cn);
            append(kRgTop, mine && !trivial && sc.tlas_root < 0, id, ht, ct);
            append(kRgFree, mine && trivial, id, hf, cf);
        } else if (kind == kRgNode) {
            uint32_t id = 0;
            Walk tr;
            if (on) id = P.list[kRgNode][(hn + (uint32_t)lane) & 255u];
            hn += take;
            cn -= take;
            const typename Walk::Stack st = stack_of(id);
            tr.have = true;
            if (on) {
                const float4 a = P.a[id], b = P.b[id];
                const uint32_t m = P.m[id];
                tr.o = mk3(a.x, a.y, a.z);
                tr.d = mk3(b.x, b.y, b.z);
                tr.hit.t = a.w;
                tr.cur = __float_as_int(b.w);
                tr.sp = (int)(m & 255u);
                tr.in_blas = (m & 256u) != 0u;
                tr.cur_inst = (m >> 9) & kInstMask;
                tr.set_level();
                tr.node_step4(sc, st, cnt);
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[1]++;
                cnt.phase[2] += take;
            }
            file_slot(on, id, tr, st);
        } else if (kind == kRgLeaf) {
            uint32_t id = 0;
            Walk tr;
            if (on) id = P.list[kRgLeaf][(hl + (uint32_t)lane) & 255u];
            hl += take;
            cl -= take;
            const typename Walk::Stack st = stack_of(id);
            tr.have = true;
            if (on) {
                const float4 a = P.a[id], b = P.b[id], c = P.c[id];
                const uint32_t m = P.m[id];
                tr.o = mk3(a.x, a.y, a.z);
                tr.d = mk3(b.x, b.y, b.z);
                tr.hit.t = a.w;
                tr.hit.u = c.x;
                tr.hit.v = c.y;
                tr.hit.tri = __float_as_uint(c.z);
                tr.hit.inst = __float_as_uint(c.w) & 0x7fffffffu;
                tr.hit.front = (__float_as_uint(c.w) >> 31) != 0u;
                tr.cur = __float_as_int(b.w);
                tr.sp = (int)(m & 255u);
                tr.in_blas = true;
                tr.cur_inst = (m >> 9) & kInstMask;
                tr.leaf_step(sc, cnt);
                P.a[id].w = tr.hit.t;
                P.c[id] = make_float4(tr.hit.u, tr.hit.v, __uint_as_float(tr.hit.tri), __uint_as_float(tr.hit.inst | (tr.hit.front ? 0x80000000u : 0u)));
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[3]++;
                cnt.phase[4] += take;
            }
            file_slot(on, id, tr, st);
        } else {
            uint32_t id = 0;
            Walk tr;
            if (on) id = P.list[kRgTop][(ht + (uint32_t)lane) & 255u];
            ht += take;
            ct -= take;
            const typename Walk::Stack st = stack_of(id);
            tr.have = true;
            if (on) {
                const uint32_t m = P.m[id];
                const size_t loc = P.loc[id];
                const float4 ro = qo[loc], rd = qd[loc];
                tr.wo = mk3(ro.x, ro.y, ro.z);
                tr.wd = mk3(rd.x, rd.y, rd.z);
                tr.cur = __float_as_int(P.b[id].w);
                tr.sp = (int)(m & 255u);
                tr.in_blas = false;
                tr.cur_inst = (m >> 9) & kInstMask;
                if (tr.cur < 0) {
                    tr.instance_step(sc, st, cnt);   // the instance's local ray, a sentinel on the stack, the BLAS root
                } else {
                    tr.o = tr.wo;                    // a TLAS record after an instance: its box tests read the world ray
                    tr.d = tr.wd;
                }
                P.a[id].x = tr.o.x;
                P.a[id].y = tr.o.y;
                P.a[id].z = tr.o.z;
                P.b[id].x = tr.d.x;
                P.b[id].y = tr.d.y;
                P.b[id].z = tr.d.z;
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[5]++;
                cnt.phase[6] += take;
            }
            file_slot(on, id, tr, st);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the hits still waiting in their slots
    for (int s = lane; s < kRgPool; s += 64)
        if (P.m[s] & kRgPending) write_hit((uint32_t)s);
    if (COUNT) flush_counters(cnt, counters);
}

