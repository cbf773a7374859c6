// jpt_kernels_wf2_pool.h -- bounces >= 1, POOLED (JPT_TRACE_REGROUP=2; VERDICT r04 task 1).  Included by jpt_kernels_wf2.hip.
//
// wf2_trace binds a ray to a lane for its whole walk, so a record step runs with the 28-39 of 64 lanes whose ray happens to
// want one and a leaf turn with 16-29: the kernel issues VALU work two thirds of the time at 37 % of its
// lanes.  Round 4's first answer (wf2_trace_rg, jpt_kernels_wf2_variants.h) kept the rays' walk state in LDS and took compacted
// lists of rays per kind of step: the lanes filled (0.50 -> 0.78) and the launch took twice as long, for two reasons it named
// and left: taking, unpacking, re-packing and filing a ray cost as much as the step it was filed for (~150 VALU + ~150 SALU
// per turn against a 137-instruction record step), and with one turn in flight per wave the chain list -> LDS state -> record
// fetch -> step -> lists lay bare (valu_issue_frac 0.36 at 2.2 waves per SIMD).  This kernel is the same idea with both
// answered:
//
//  * A turn is PREPARED one turn ahead.  Before the wave computes turn k it takes turn k + 1's rays off the lists (they are
//    not turn k's: those are checked out), reads their state from LDS and asks for their records -- the 64-byte WideNodeQ, the
//    leaf's two WideTri, the instance record and the world ray -- so the fetch of turn k + 1 is in flight while turn k's
//    ~200 instructions issue.  When no list is long enough to be worth a turn yet (the pool is draining), the wave waits for
//    turn k's rays to be filed and chooses then.
//  * The per-ray state a turn moves is what the step needs and no more: a record step reads two 16-byte words (level ray
//    origin + closest distance, direction + current record) and one packed word (stack height, level, instance) and writes
//    back ONE 8-byte pair (record, packed word) -- the level ray changes only at an instance entry, the closest distance
//    only at a leaf.  The closest hit's u, v, triangle and instance never sit in LDS: a leaf turn that accepts a triangle
//    writes the ray's hit record (hit_a / hit_b, where wf2_shade reads it) on the spot, and a finished ray that never hit
//    writes its miss then; 40 + 4 x kPoolStack bytes of LDS per ray instead of 88.
//  * Filing is one ballot and one prefix count per list (v_mbcnt pair with the list's base folded in, one ds_write_b8 under
//    the ballot's exec mask), the pops that precede it are the walk's own (Traversal::pop).
//
// Per ray the sequence of steps -- the functions of jpt_trace_core.h on the same records, the same stack discipline, the same
// tie flag -- is exactly wf2_trace's, so the hits are bit-identical; only which rays share a wave-level instruction changes.
// A wave (= a block of 64 threads) owns its pool; several waves share a queue through a cursor in global memory (one atomic
// per 64 rays, reserved one refill ahead).
#pragma once

#ifndef JPT_POOL_RAYS
#define JPT_POOL_RAYS 192
#endif
#ifndef JPT_POOL_STACK
#define JPT_POOL_STACK 6
#endif
#ifndef JPT_POOL_PREFETCH
#define JPT_POOL_PREFETCH 1   // 0: every turn is chosen after the previous one is filed (A/B: what the prefetch buys)
#endif
constexpr int kPoolRays = JPT_POOL_RAYS;       // rays in flight per wave (slot ids are bytes)
constexpr int kPoolStack = JPT_POOL_STACK;     // stack entries per ray in LDS ...
constexpr int kPoolSpill = 64 - JPT_POOL_STACK;   // ... and past them in global memory: 64 in all (main.glsl:272,307)
static_assert(kPoolRays >= 128 && kPoolRays <= 256 && kPoolRays % 64 == 0, "pool slots are addressed by bytes; a refill takes 64");
enum { kPoolNode = 0, kPoolLeaf = 1, kPoolTop = 2, kPoolFree = 3, kPoolNone = 4 };

struct PoolLds {
    float4 a[kPoolRays];                    // level ray origin.xyz, closest distance so far (1e9: none)
    float4 b[kPoolRays];                    // level ray direction.xyz, -
    uint2 cm[kPoolRays];                    // x: current record (child reference bits); y: sp (0..7) | in_blas (8) | cur_inst (9..23)
    uint32_t loc[kPoolRays];                // the ray's queue entry (world ray in, hit out)
    int32_t stack[kPoolStack * kPoolRays];  // [entry][slot]
    uint8_t ring[4][256];                   // rings of slot ids: NODE, LEAF, TOP, FREE
};

size_t wf2_pool_spill_bytes(uint32_t blocks) { return (size_t)blocks * kPoolRays * kPoolSpill * sizeof(int32_t); }
constexpr uint32_t kPoolMaxBlocks = 4096u;   // grid cap of a pooled launch (the spill area is sized for it)

// a turn, chosen and (its loads) asked for: kind / take are wave-uniform
struct PoolTurn {
    int kind;
    uint32_t take;        // lanes 0 .. take - 1 hold a ray
    uint32_t id;          // pool slot
    float4 a, b;          // its state
    uint2 cm;
    uint32_t loc;
    float4 r0, r1, r2, r3, r4, r5;   // records: NODE r0..r3 = the WideNodeQ; LEAF r0..r2 / r3..r5 = the leaf's first / second WideTri;
                                     // TOP r0..r3 = the WideInstance, r4 / r5 = the world ray's origin / direction; FREE r4 / r5 = the new ray
};

// A value nobody reads, made by no instruction: the fields of a turn that its kind of step does not use.  (Left as they were
// they would be merged with the previous turn's values at every join: forty v_mov per turn.)
__device__ __forceinline__ float4 nobody_reads4()
{
    typedef float v4f __attribute__((ext_vector_type(4)));   // (one 128-bit register tuple, like the loads it stands in for)
    v4f v;
    asm("" : "=v"(v));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 nobody_reads2()
{
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    v2u v;
    asm("" : "=v"(v));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t nobody_reads()
{
    uint32_t v;
    asm("" : "=v"(v));
    return v;
}

// The wave's view of its pool.  (A struct with force-inlined members, not lambdas: closures that capture other closures by
// reference kept every captured variable -- list heads, cursors, the kernel's argument structs -- in scratch memory.)
template <bool COUNT>
struct PoolWave {
    using Walk = Traversal<COUNT, true>;
    PoolLds* __restrict__ P;
    int lane;
    // the queue: `chain` consecutive segments, shared with the other waves of the queue through `cursor`
    uint32_t seg0, seg_cap, n, end0, end1, end2;
    uint32_t* __restrict__ cursor;
    const float4* __restrict__ qo;
    const float4* __restrict__ qd;
    float4* __restrict__ hit_a;
    uint32_t* __restrict__ hit_b;
    int32_t* __restrict__ spill_base;
    const char* __restrict__ nodes;
    const char* __restrict__ tris;
    const char* __restrict__ insts;
    int32_t tlas_root;
    uint32_t n_instances;
    uint32_t hn, hl, ht, hf;   // list heads (free-running; the rings hold 256)
    uint32_t cn, cl, ct, cf;   // list sizes
    uint32_t next_start;       // 64 queue entries are reserved one refill ahead
    bool exhausted;
    DevCounters cnt;

    // one list gets the slots of the lanes in `pred`: ballot, prefix count with the list's end folded in, one byte store each
    __device__ __forceinline__ void append(int kind, bool pred, uint32_t id, uint32_t head, uint32_t& count)
    {
        const unsigned long long mk = __ballot(pred);
        if (mk) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, head + count));
            if (pred) P->ring[kind][pos & 255u] = (uint8_t)id;
            count += (uint32_t)__popcll(mk);
        }
    }
    __device__ __forceinline__ typename Walk::Stack stack_of(uint32_t id) const
    {
        return typename Walk::Stack{&P->stack[id], spill_base + (size_t)id * kPoolSpill, kPoolRays, kPoolStack, kPoolSpill};
    }

    // ---- choosing a turn: a refill when 64 slots are free, else a full chunk if a list has one, else the longest list ---------
    // `at_least`: the fewest rays worth a turn NOW (a turn prepared ahead cannot take the rays of the turn in flight: when the
    // lists are short because of that, waiting for them to be filed gives a fuller turn)
    __device__ __forceinline__ bool choose(PoolTurn& t, uint32_t at_least)
    {
        if (!exhausted && cf >= 64u) {
            const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_start);
            const uint32_t avail = start < n ? (n - start < 64u ? n - start : 64u) : 0u;
            // (`exhausted` set BEFORE the one-lane branch: as `if (..) exhausted = true; else if (lane == 0) ..` it became a value
            // merged at the join of a divergent branch, i.e. divergent to the compiler, and with it every list head and count: VGPRs,
            // exec-mask branches)
            const bool last = start + 64u >= n;
            exhausted = last;
            if (!last && lane == 0) next_start = atomicAdd(cursor, 64u);
            if (avail != 0u) {   // (0: the queue ended exactly at the last refill -- on to the lists)
                t.kind = kPoolFree;
                t.take = avail;
                // (every lane: a lane without a ray takes the first of the reserved entries again -- no field of the turn depends on
                // what the previous turn left in it)
                const uint32_t mine = (uint32_t)lane < avail ? (uint32_t)lane : 0u;
                t.id = P->ring[kPoolFree][(hf + mine) & 255u];
                const uint32_t idx = start + mine;
                // which of the chained segments holds entry idx
                uint32_t k = 0, first = 0;
                if (idx >= end0) k = 1u, first = end0;
                if (idx >= end1) k = 2u, first = end1;
                if (idx >= end2) k = 3u, first = end2;
                t.loc = (seg0 + k) * seg_cap + (idx - first);
                t.r0 = qo[t.loc];
                t.r1 = qd[t.loc];
                t.r2 = ld4(nodes), t.r3 = ld4(nodes + 16);   // (every turn asks for FOUR 16-byte words: see take())
                t.a = t.b = t.r4 = t.r5 = nobody_reads4();
                t.cm = nobody_reads2();
                hf += avail;
                cf -= avail;
                return true;
            }
        }
        const uint32_t best = cn >= cl ? (cn >= ct ? cn : ct) : (cl >= ct ? cl : ct);
        if (best == 0u || best < at_least) {
            // (no turn, no loads: the wait makes this path agree with the others about what is outstanding afterwards -- nothing that
            // a later turn could be kept waiting for; the wave has nothing else to do here anyway)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            t.kind = kPoolNone;
            t.take = 0u;
            t.id = t.loc = nobody_reads();
            t.a = t.b = t.r0 = t.r1 = t.r2 = t.r3 = t.r4 = t.r5 = nobody_reads4();
            t.cm = nobody_reads2();
            return false;
        }
        // (one copy of the take per list, each naming its own head and count: a run-time choice between them compiles to a
        // select between ADDRESSES, which keeps this whole struct in scratch memory)
        if (cn >= 64u || (cl < 64u && ct < 64u && best == cn)) take<kPoolNode>(t, hn, cn);
        else if (cl >= 64u || (ct < 64u && best == cl)) take<kPoolLeaf>(t, hl, cl);
        else take<kPoolTop>(t, ht, ct);
        return true;
    }
    // up to 64 rays off one list: their slots, their state, and the loads of the records their step will read
    template <int KIND>
    __device__ __forceinline__ void take(PoolTurn& t, uint32_t& head, uint32_t& count)
    {
        const uint32_t take = count < 64u ? count : 64u;
        t.kind = KIND;
        t.take = take;
        // (every lane loads: a lane without a ray repeats lane 0's -- no field of the turn depends on what the previous turn left in it)
        const uint32_t mine = (uint32_t)lane < take ? (uint32_t)lane : 0u;
        t.id = P->ring[KIND][(head + mine) & 255u];
        head += take;
        count -= take;
        t.a = P->a[t.id];
        t.b = P->b[t.id];
        t.cm = P->cm[t.id];
        const int32_t cur = (int32_t)t.cm.x;
        // EXACTLY FOUR 16-byte global loads, whatever the kind.  The memory counter of this chip is in order: the wait at the start
        // of a turn can leave the NEXT turn's loads in flight only if the compiler knows how many of them there are -- on every
        // path alike; with 2 / 4 / 6 loads depending on the kind it waited for everything (vmcnt(0)) and the prefetch bought
        // nothing (profiles/r05/r05b_*: no difference between preparing ahead and not).  What does not fit in four is asked for at
        // the start of the turn itself: the second triangle's edges, the instance's third row and root.
        if (KIND == kPoolNode) {
            const char* p = nodes + ((size_t)(uint32_t)cur << 6);
            t.r0 = ld4(p), t.r1 = ld4(p + 16), t.r2 = ld4(p + 32), t.r3 = ld4(p + 48);
            t.r4 = t.r5 = nobody_reads4();
            t.loc = nobody_reads();
        } else if (KIND == kPoolLeaf) {
            const uint32_t bits = (uint32_t)~cur;
            const uint32_t first = bits & kLeafFirstMask, second = first + ((bits >> kLeafCountShift) != 0u ? 1u : 0u);
            const char* p = tris + (size_t)first * sizeof(WideTri);
            const char* q = tris + (size_t)second * sizeof(WideTri);   // (a leaf of one triangle asks for it twice: no read past the array)
            t.r0 = ld4(p), t.r1 = ld4(p + 16), t.r2 = ld4(p + 32);
            t.r3 = ld4(q);
            t.r4 = t.r5 = nobody_reads4();
            t.loc = P->loc[t.id];
        } else {
            t.loc = P->loc[t.id];
            t.r4 = qo[t.loc];
            t.r5 = qd[t.loc];
            // (a TLAS record after an instance was left reads no instance: record 0 stands in)
            const char* p = insts + (size_t)(cur < 0 ? (uint32_t)~cur : 0u) * sizeof(WideInstance);
            t.r0 = ld4(p), t.r1 = ld4(p + 16);
            t.r2 = t.r3 = nobody_reads4();
        }
    }

    // ---- after a step: the next record off the ray's stack if the step left none, then the slot goes to the list of what the
    // ray wants next (pop_next of jpt_trace_core.h; a ray that left an instance wants the world ray back: TOP) ------------------
    __device__ __forceinline__ void file_slot(bool on, uint32_t id, Walk& tr, const typename Walk::Stack& st, float t_now)
    {
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
        if (on) P->cm[id] = make_uint2((uint32_t)tr.cur, (uint32_t)tr.sp | (tr.in_blas ? 256u : 0u) | (tr.cur_inst << 9));
        const bool go = on && !fin;
        const bool to_top = go && !tr.in_blas && (tr.cur < 0 || left_instance);
        const bool to_node = go && tr.cur >= 0 && !to_top;
        const bool to_leaf = go && tr.cur < 0 && tr.in_blas;
        append(kPoolNode, to_node, id, hn, cn);
        append(kPoolLeaf, to_leaf, id, hl, cl);
        append(kPoolTop, to_top, id, ht, ct);
        append(kPoolFree, on && fin, id, hf, cf);
        // a finished ray that never accepted a triangle: its miss (a hit was written when it was accepted)
        if (on && fin && !(t_now < 1e9f)) hit_a[P->loc[id]] = make_float4(1e9f, 0.0f, 0.0f, 0.0f);
    }

    // ---- one turn ---------------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void run(const PoolTurn& t)
    {
        const bool on = (uint32_t)lane < t.take;
        const uint32_t id = t.id;
        const typename Walk::Stack st = stack_of(id);
        Walk tr;
        tr.have = true;
        if (t.kind == kPoolFree) {
            // refill: the reserved queue entries start their walk at the TLAS root with the world ray; closest distance 1e9 (main.glsl:354)
            const bool trivial = n_instances == 0u;
            if (on) {
                P->a[id] = make_float4(t.r0.x, t.r0.y, t.r0.z, 1e9f);
                P->b[id] = make_float4(t.r1.x, t.r1.y, t.r1.z, 0.0f);
                P->cm[id] = make_uint2((uint32_t)tlas_root, 0u);
                P->loc[id] = t.loc;
                if (trivial) hit_a[t.loc] = make_float4(1e9f, 0.0f, 0.0f, 0.0f);
            }
            append(kPoolNode, on && !trivial && tlas_root >= 0, id, hn, cn);
            append(kPoolTop, on && !trivial && tlas_root < 0, id, ht, ct);
            append(kPoolFree, on && trivial, id, hf, cf);
            asm volatile("" ::"v"(t.r2.x), "v"(t.r3.x));   // (the two stand-in loads are part of the count: not to be optimised away)
            return;
        }
        const uint32_t m = t.cm.y;
        tr.o = mk3(t.a.x, t.a.y, t.a.z);
        tr.d = mk3(t.b.x, t.b.y, t.b.z);
        tr.hit.t = t.a.w;
        tr.cur = (int32_t)t.cm.x;
        tr.sp = (int)(m & 255u);
        tr.in_blas = (m & 256u) != 0u;
        tr.cur_inst = (m >> 9) & kInstMask;
        if (t.kind == kPoolNode) {
            if (on) {
                tr.set_level();
                tr.node_step4_rec(t.r0, t.r1, t.r2, t.r3, st, cnt);
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[1]++;
                cnt.phase[2] += t.take;
            }
            file_slot(on, id, tr, st, t.a.w);
        } else if (t.kind == kPoolLeaf) {
            float t_now = t.a.w;
            if (on) {
                // Traversal::leaf_step's four-child form (intersectTriangle, main.glsl:224-257, the early returns folded into one
                // predicate), on the leaf's records as prefetched; the accepted triangle goes straight to the ray's hit record
                const uint32_t bits = (uint32_t)~tr.cur;
                const uint32_t first = bits & kLeafFirstMask, count = (bits >> kLeafCountShift) + 1u;
                bool accepted = false, inst_known = false;
                float hu = 0.0f, hv = 0.0f;
                uint32_t htri = 0u, hinst = 0u;
                bool hfront = false;
                // (one triangle test, by value: choosing the record through the loop counter made the compiler choose between the
                // ADDRESSES of the turn's fields and keep the turn in scratch memory)
                auto test = [&](const float4 q0, const float4 q1, const float4 q2, const uint32_t ti) __attribute__((always_inline)) {
                    if (COUNT) cnt.tri_tests++;
                    const f3 v0 = mk3(q0.x, q0.y, q0.z), edge1 = mk3(q1.x, q1.y, q1.z), edge2 = mk3(q2.x, q2.y, q2.z);
                    const f3 pvec = cross3(tr.d, edge2);
                    const float det = dot3(edge1, pvec);
                    const float invDet = 1.0f / det;
                    const f3 tvec = tr.o - v0;
                    const float u = dot3(tvec, pvec) * invDet;
                    const f3 qvec = cross3(tvec, edge1);
                    const float v = dot3(tr.d, qvec) * invDet;
                    const float tt = dot3(edge2, qvec) * invDet;
                    const bool out = (__builtin_fabsf(det) < 1e-5f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (tt < 0.0f) | (tt > t_now);
                    const float facing = dot3(mk3(q0.w, q1.w, q2.w), tr.d);   // the record carries cross(edge1, edge2)
                    if (!out) {
                        // hitInfo.blas moves only on a strictly smaller distance (main.glsl:324-327); an exact tie keeps the instance
                        // of the hit in hand and is flagged (kHitTied) -- the hit in hand is this turn's, or the one in the ray's
                        // hit record (or none at all: t_now is still 1e9)
                        const bool closer = tt < t_now;
                        if (!closer && !inst_known) {
                            hinst = 0u;
                            if (t_now < 1e9f) {   // (an exact tie with a hit of an earlier turn: rare -- the record as this wave last wrote it)
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                                hinst = __hip_atomic_load(&hit_b[t.loc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fffffffu;
                            }
                        }
                        hinst = ((closer ? tr.cur_inst : hinst) & kInstMask) | (tr.cur_inst << kInstBits) | (closer ? 0u : kHitTied);
                        inst_known = true;
                        t_now = tt;
                        hu = u;
                        hv = v;
                        htri = ti;
                        hfront = facing > 0.0f;
                        accepted = true;
                    }
                };
                // (the second triangle's two edges: asked for now, used after the first triangle's test)
                const char* q = tris + (size_t)(first + (count > 1u ? 1u : 0u)) * sizeof(WideTri);
                const float4 s1 = ld4(q + 16), s2 = ld4(q + 32);
                test(t.r0, t.r1, t.r2, first);
                if (count > 1u) test(t.r3, s1, s2, first + 1u);
                for (uint32_t i = 2; i < count; i++) {   // (leaves of more than two triangles: JPT_MAX_LEAF > 2)
                    const char* p = tris + (size_t)(first + i) * sizeof(WideTri);
                    test(ld4(p), ld4(p + 16), ld4(p + 32), first + i);
                }
                if (accepted) {
                    P->a[id].w = t_now;
                    hit_a[t.loc] = make_float4(t_now, hu, hv, __uint_as_float(htri));
                    hit_b[t.loc] = hinst | (hfront ? 0x80000000u : 0u);
                }
                tr.have = false;
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[3]++;
                cnt.phase[4] += t.take;
            }
            file_slot(on, id, tr, st, t_now);
        } else {
            if (on) {
                tr.wo = mk3(t.r4.x, t.r4.y, t.r4.z);
                tr.wd = mk3(t.r5.x, t.r5.y, t.r5.z);
                tr.in_blas = false;
                if (tr.cur < 0) {
                    // Traversal::instance_step on the prefetched instance record: the instance's local ray (main.glsl:319-320), a
                    // sentinel on the stack, the BLAS root
                    tr.cur_inst = (uint32_t)~tr.cur;
                    const char* ip = insts + (size_t)tr.cur_inst * sizeof(WideInstance);
                    const float4 m0 = t.r0, m1 = t.r1, m2 = ld4(ip + 32);   // (third row and root: asked for now -- a handful of instances are L1 hits)
                    const int32_t root = *reinterpret_cast<const int32_t*>(ip + 48);
                    if (COUNT) cnt.inst_visits++;
                    tr.o = mk3(m0.x * tr.wo.x + m0.w * tr.wo.y + m1.z * tr.wo.z + m2.y, m0.y * tr.wo.x + m1.x * tr.wo.y + m1.w * tr.wo.z + m2.z,
                               m0.z * tr.wo.x + m1.y * tr.wo.y + m2.x * tr.wo.z + m2.w);
                    tr.d = mk3(m0.x * tr.wd.x + m0.w * tr.wd.y + m1.z * tr.wd.z, m0.y * tr.wd.x + m1.x * tr.wd.y + m1.w * tr.wd.z,
                               m0.z * tr.wd.x + m1.y * tr.wd.y + m2.x * tr.wd.z);
                    tr.in_blas = true;
                    tr.push(st, kSentinel);
                    tr.cur = root;
                } else {
                    tr.o = tr.wo;   // a TLAS record after an instance: its box tests read the world ray
                    tr.d = tr.wd;
                }
                P->a[id] = make_float4(tr.o.x, tr.o.y, tr.o.z, t.a.w);
                P->b[id] = make_float4(tr.d.x, tr.d.y, tr.d.z, 0.0f);
            }
            if (COUNT && lane == 0) {
                cnt.phase[0]++;
                cnt.phase[5]++;
                cnt.phase[6] += t.take;
            }
            file_slot(on, id, tr, st, t.a.w);
        }
    }
};

template <bool COUNT>
__global__ __launch_bounds__(64) void wf2_trace_pool(WideSceneDev sc, Wf2Buffers wb, Wf2Dims dm, int bounce, int chain, int waves_per_queue,
                                                     int min_prefetch, DevCounters* __restrict__ counters)
{
    __shared__ PoolLds lds;
    PoolWave<COUNT> w;
    w.P = &lds;
    w.lane = threadIdx.x;
    const uint32_t wave_in_queue = blockIdx.x % (uint32_t)waves_per_queue;
    w.seg0 = (blockIdx.x / (uint32_t)waves_per_queue) * (uint32_t)chain;
    constexpr int kPoolChain = 4;   // (PoolWave::choose finds an entry's segment among four: the launcher chains at most that many)
    uint32_t end[kPoolChain];
    uint32_t n = 0;
    for (int k = 0; k < kPoolChain; k++) {
        if (k < chain && w.seg0 + (uint32_t)k < kSegments) n += wb.qcount[(size_t)bounce * kSegments + w.seg0 + (uint32_t)k];
        end[k] = n;
    }
    // a queue of n rays is worth n / 1024 waves (at least one): a pool that is not refilled a few times runs mostly half empty
    if (n == 0 || wave_in_queue * 1024u >= n) return;
    w.n = n;
    w.end0 = end[0], w.end1 = end[1], w.end2 = end[2];
    w.seg_cap = dm.seg_cap;
    w.cursor = wb.rg_cursor + (size_t)bounce * kSegments + w.seg0;
    w.qo = wb.ray_o[bounce & 1];
    w.qd = wb.ray_d[bounce & 1];
    w.hit_a = wb.hit_a;
    w.hit_b = wb.hit_b;
    w.spill_base = wb.rg_spill + (size_t)blockIdx.x * kPoolRays * kPoolSpill;
    w.nodes = reinterpret_cast<const char*>(sc.nodesq);
    w.tris = reinterpret_cast<const char*>(sc.tris);
    w.insts = reinterpret_cast<const char*>(sc.instances);
    w.tlas_root = sc.tlas_root;
    w.n_instances = sc.n_instances;
    w.cnt = DevCounters{};
    for (int s = w.lane; s < kPoolRays; s += 64) lds.ring[kPoolFree][s] = (uint8_t)s;
    w.hn = w.hl = w.ht = w.hf = 0;
    w.cn = w.cl = w.ct = 0;
    w.cf = kPoolRays;
    w.next_start = 0;
    if (w.lane == 0) w.next_start = atomicAdd(w.cursor, 64u);
    w.exhausted = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- the loop: turn k + 1 is chosen and its records asked for before turn k computes.  Two turns `a` and `b` take turns
    // (unrolled by two: with one "current" and one "next" record set the hand-over at the end of every turn was a copy of forty
    // registers) -----------------------------------------------------------------------------------------------------------------

    const uint32_t at_least = JPT_POOL_PREFETCH ? (uint32_t)min_prefetch : 0xffffffffu;
    auto wave_sync = [] {   // the turn's LDS writes before the next turn's reads (one wave: no barrier instruction)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // (ONE choose() and one run() per record set and iteration, and a choose() that fails leaves nothing of the previous turn in
    // the set: otherwise the compiler merges the set's old and new values at every join -- forty v_mov per turn)
    PoolTurn b;
    bool pend_b = false;
    for (;;) {
        PoolTurn a;
        const bool pend_a = w.choose(a, pend_b ? at_least : 1u);
        if (pend_b) {
            w.run(b);
            wave_sync();
        } else if (!pend_a) {
            break;   // nothing in flight, nothing listed, no refill left: the queue is done
        }
        pend_b = w.choose(b, pend_a ? at_least : 1u);
        if (pend_a) {
            w.run(a);
            wave_sync();
        } else if (!pend_b) {
            break;
        }
    }
    if (COUNT) flush_counters(w.cnt, counters);
}
