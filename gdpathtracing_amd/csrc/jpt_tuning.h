// jpt_tuning.h -- the library's environment switches, read ONCE per process (first use, normally the first jpt_create).
// They exist for tuning runs and tests; the defaults are the measured optima.  Nothing on the
// render path calls getenv.
#pragma once

#include <cstdlib>

namespace jpt {

struct Tuning {
    bool sky_cull = true;          // JPT_SKY_CULL=0: trace every primary ray
    long workspace_budget_mb = 24576;  // JPT_WORKSPACE_BUDGET_MB: frames in flight per launch set
    bool pipelining = true;        // JPT_PIPELINE=0: asynchronous renders run one after another
    int pipe_slots = 0;            // JPT_PIPE_SLOTS=2..8: renders in flight (0: the library's rule -- 4, or 6 where six slot streams run side by side)
    bool six_slots = true;         // JPT_SIX_SLOTS=0: never more than four by rule
    bool acc_on_slot = true;       // JPT_ACC_ON_SLOT=0: accumulation kernels on the context's stream
    int slot_prio = 3;             // JPT_SLOT_PRIO: priority of the pipeline slots' streams -- 0 all normal, 1 dealt over the device's
                                   // priority levels, 2 two high + two low, 3 (default) all high, 4 all low (jpt_capi.hip, ensure_pipe_slot)
    int bvh_width = 4;             // JPT_BVH_WIDTH=2: two-child records on the native tree
    int run_shift = -1;            // JPT_RUN_SHIFT: tiles dealt to a segment in runs of 2^n (-1: the library's rule)
    int groups = 0;                // JPT_GROUPS=1..4: frame groups of a blocking render (0: the library's rule)
    int refill_idle = 24;          // JPT_REFILL_IDLE: a wave refills when this many lanes are idle (20..32 x 20..32 swept: a 1 % plateau)
    int primary_refill_idle = -1;  // JPT_PRIMARY_REFILL_IDLE: ... in the primary launch (-1: as refill_idle)
    int node_min_lanes = 24;       // JPT_NODE_MIN_LANES: leave the record loop below this many descending lanes
    int leaf_min_lanes = 16;       // JPT_LEAF_MIN_LANES / JPT_INST_MIN_LANES: a leaf / instance phase with fewer takers waits a round
    int inst_min_lanes = 12;       // (while the wave has other work); 1 = every round, as before
    int phase_frac16 = 4;          // JPT_PHASE_FRAC16: ... capped at this many sixteenths of the wave's active rays
    int trace_chain = 0;           // JPT_TRACE_CHAIN=1..4: segments per block of the tracing launches (0: the library's rule)
    int max_leaf = 2;              // JPT_MAX_LEAF: native builder leaf size
    bool upload_as_given = false;  // JPT_UPLOAD_WALK=given: reference-layout uploads are walked node for node as uploaded (audits)
    bool exact_shadow = true;      // JPT_EXACT_SHADOW=0: no copy of the reference's trees beside a native scene (set-aside paths are
                                   // finished on the native tree with the reach tests inside the walk; exact ties not re-traced)
    long set_aside_cap = -1;       // JPT_SET_ASIDE_CAP: records of the set-aside buffer (-1: the library's rule; tests force 0)
    int reach = 2;                 // JPT_REACH=0: ignore the reach records; 1: check but never redo (timing experiments only)
    bool shade_last = true;        // JPT_SHADE_LAST=0: the final shading launch uses the general kernel (A/B)
    int lone_async = 1;            // JPT_LONE_ASYNC=0: queued renders that find the pipeline empty are never launched like blocking ones
    int node_order = 0;            // JPT_NODE_ORDER=1: the four-child records of siblings next to each other (0: depth first)
    int primary_samples = -1;      // JPT_PRIMARY_SAMPLES=0/1: a wave of the primary launch takes one frame's sample of an 8 x 8 tile / every frame's
                                   // sample of 64 / n_frames pixels of a tile (-1: the library's rule)
    int tail = -1;                 // JPT_TAIL: the waves of a tracing launch walk their last, long rays themselves, all lanes on one ray
                                   // (coop_walk_call): -1 on scenes of >= 200 000 triangles, 0 never, 1 on every scene
    int tail_rounds = 128;         // JPT_TAIL_ROUNDS: ... from this many rounds after the block's queue ran dry
    int tail_lanes = 8;            // JPT_TAIL_LANES: ... once a wave is down to this many rays
    int collapse = 3;              // JPT_COLLAPSE: two-child records merged into four-child ones by the least-area plan (jpt_builder.cpp, CollapsePlan) -- 0 greedily (the largest box first: rounds 1-4), 1 the TLAS, 2 the meshes' trees, 3 both
    int instance_boxes = 1024;     // JPT_INSTANCE_BOXES: a native scene's instance boxes bound up to this many boxes of the mesh's tree, transformed one by one
                                   // (1: the root box's corners, as the reference's rule and rounds 1-4)
};

inline const Tuning& tuning()
{
    static const Tuning t = [] {
        Tuning v;
        auto geti = [](const char* name, int dflt) {
            const char* e = std::getenv(name);
            return e ? std::atoi(e) : dflt;
        };
        v.sky_cull = geti("JPT_SKY_CULL", 1) != 0;
        if (const char* e = std::getenv("JPT_WORKSPACE_BUDGET_MB")) v.workspace_budget_mb = std::atol(e);
        v.pipelining = geti("JPT_PIPELINE", 1) != 0;
        v.pipe_slots = geti("JPT_PIPE_SLOTS", 0);
        v.six_slots = geti("JPT_SIX_SLOTS", 1) != 0;
        v.acc_on_slot = geti("JPT_ACC_ON_SLOT", 1) != 0;
        v.slot_prio = geti("JPT_SLOT_PRIO", 3);
        v.bvh_width = geti("JPT_BVH_WIDTH", 4) == 2 ? 2 : 4;
        v.run_shift = geti("JPT_RUN_SHIFT", -1);
        if (v.run_shift > 8) v.run_shift = 8;
        v.groups = geti("JPT_GROUPS", 0);
        v.refill_idle = geti("JPT_REFILL_IDLE", 24);
        v.primary_refill_idle = geti("JPT_PRIMARY_REFILL_IDLE", -1);
        if (v.primary_refill_idle < 0) v.primary_refill_idle = v.refill_idle;
        v.node_min_lanes = geti("JPT_NODE_MIN_LANES", 24);
        v.leaf_min_lanes = geti("JPT_LEAF_MIN_LANES", 16);
        v.inst_min_lanes = geti("JPT_INST_MIN_LANES", 12);
        v.phase_frac16 = geti("JPT_PHASE_FRAC16", 4);
        v.trace_chain = geti("JPT_TRACE_CHAIN", 0);
        v.max_leaf = geti("JPT_MAX_LEAF", 2);
        v.reach = geti("JPT_REACH", 2);
        v.shade_last = geti("JPT_SHADE_LAST", 1) != 0;
        v.lone_async = geti("JPT_LONE_ASYNC", 1);
        v.node_order = geti("JPT_NODE_ORDER", 0);
        v.primary_samples = geti("JPT_PRIMARY_SAMPLES", -1);
        v.tail = geti("JPT_TAIL", -1);
        v.tail_rounds = geti("JPT_TAIL_ROUNDS", 128);
        v.tail_lanes = geti("JPT_TAIL_LANES", 8);
        v.collapse = geti("JPT_COLLAPSE", 3);
        v.instance_boxes = geti("JPT_INSTANCE_BOXES", 1024);
        v.exact_shadow = geti("JPT_EXACT_SHADOW", 1) != 0;
        if (const char* e = std::getenv("JPT_SET_ASIDE_CAP")) v.set_aside_cap = std::atol(e);
        if (const char* e = std::getenv("JPT_UPLOAD_WALK")) v.upload_as_given = e[0] == 'g' || e[0] == 'G';
        if (v.max_leaf < 1) v.max_leaf = 1;
        if (v.max_leaf > 16) v.max_leaf = 16;
        return v;
    }();
    return t;
}

}  // namespace jpt
