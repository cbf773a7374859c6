// jpt_tuning.h -- the library's environment switches (the table in include/jpt.h), read ONCE per process (first use, normally
// the first jpt_create), and the constants of the wavefront kernels' scheduling.  Nothing on the render path calls getenv.
// The scheduling constants are the measured optima of rounds 1-5 (the sweeps: LAB_NOTEBOOK.md); they were environment
// switches while they were being swept and are plain constants now.
#pragma once

#include <cstdlib>

namespace jpt {

struct Tuning {
    // ---- environment switches: what a host, a test or a profiling run needs --------------------------------------------------
    bool sky_cull = true;              // JPT_SKY_CULL=0: trace every primary ray (audits)
    long workspace_budget_mb = 24576;  // JPT_WORKSPACE_BUDGET_MB: most MiB one render's workspace may take (jpt_set_memory_policy overrides)
    bool pipelining = true;            // JPT_PIPELINE=0: asynchronous renders run one after another (per-kernel profiling)
    int pipe_slots = 0;                // JPT_PIPE_SLOTS=2..8: renders in flight (0: the library's rule -- 4, or 6 where six slot streams run side by side)
    int groups = 0;                    // JPT_GROUPS=1..4: frame groups of a blocking render (0: the library's rule; 1 for per-kernel profiling)
    bool upload_as_given = false;      // JPT_UPLOAD_WALK=given: reference-layout uploads are walked node for node as uploaded (audits)
    long set_aside_cap = -1;           // JPT_SET_ASIDE_CAP: records of the set-aside buffer (-1: the library's rule; tests force 0)
    int tail = -1;                     // JPT_TAIL: the waves of a tracing launch walk their last, long rays themselves, all lanes on one ray
                                       // (coop_walk_call): -1 on scenes of >= 200 000 triangles, 0 never, 1 on every scene
    int tail_rounds = 128;             // JPT_TAIL_ROUNDS: ... from this many rounds after the block's queue ran dry
    int tail_lanes = 8;                // JPT_TAIL_LANES: ... once a wave is down to this many rays

    // ---- constants ---------------------------------------------------------------------------------------------------------------
    static constexpr int refill_idle = 24;      // a wave refills when this many lanes are idle (20..32: a 1 % plateau)
    static constexpr int node_min_lanes = 24;   // leave the record loop below this many descending lanes
    static constexpr int leaf_min_lanes = 16;   // a leaf / instance phase with fewer takers waits a round (while the wave has other work) ...
    static constexpr int inst_min_lanes = 12;
    static constexpr int phase_frac16 = 4;      // ... capped at this many sixteenths of the wave's active rays
    static constexpr int max_leaf = 2;          // native builder: most triangles a leaf keeps without a split that pays
    static constexpr int instance_boxes = 1024; // a native scene's instance boxes bound up to this many boxes of the mesh's tree
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
        v.groups = geti("JPT_GROUPS", 0);
        v.tail = geti("JPT_TAIL", -1);
        v.tail_rounds = geti("JPT_TAIL_ROUNDS", 128);
        v.tail_lanes = geti("JPT_TAIL_LANES", 8);
        if (const char* e = std::getenv("JPT_SET_ASIDE_CAP")) v.set_aside_cap = std::atol(e);
        if (const char* e = std::getenv("JPT_UPLOAD_WALK")) v.upload_as_given = e[0] == 'g' || e[0] == 'G';
        return v;
    }();
    return t;
}

}  // namespace jpt
