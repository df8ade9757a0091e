// switches.hpp -- the library's tuning / debugging switches (internal to libmi355rt).
//
// The MI355RT_* environment variables are read ONCE, the first time any entry point asks (getenv on the frame path is a
// dozen calls per frame and races with setenv in a multi-threaded host such as the JVM behind the JNI shim); after that
// only mi355rt_debug_switch (include/mi355rt.h) changes them, which is what the tests use to run one frame in several forms.
#pragma once

namespace mi355rt {

struct Switches {
   int builder_host = 0;           // MI355RT_BUILDER=host: never build a hierarchy on the GPU
   int build_timing = 0;           // MI355RT_BUILD_TIMING: print scene-creation phases to stderr
   int fuse_shadow = -1;           // MI355RT_FUSE_SHADOW=0/1: force the separate / fused shadow pipeline (-1: the library decides)
   int pass_slots_log2 = 0;        // MI355RT_PASS_SLOTS_LOG2: force the size of a pass (0: as large as memory allows)
   int big_leaves = -1;            // MI355RT_BIG_LEAVES=0/1: force the two-sided / burst form of the walker (-1: by the scene's leaves)
   int blocks_per_cu = 0;          // MI355RT_BLOCKS_PER_CU: workgroups per CU of the closest-hit launches (0: fitted to the pass)
   int blocks_per_cu_shadow = 0;   // MI355RT_BLOCKS_PER_CU_SHADOW
   int trim = 1;                   // MI355RT_TRIM=0: launches are not trimmed on the device to the rays actually queued
   int trim_primary_x2 = 5;        // MI355RT_TRIM_PRIMARY_X2: rays per lane (x2) a closest-hit launch is trimmed to
   int trim_shadow_x2 = 0;         // MI355RT_TRIM_SHADOW_X2 (0: 12, or 3 for scenes of big leaves)
   int spread_sl = -1;             // MI355RT_SPREAD_SL / _BL: columns / block (log2) of the spread dealing of short passes
   int spread_bl = -1;
   int spread_primary_max = 3000000;   // MI355RT_SPREAD_PRIMARY_MAX: passes of fewer samples are dealt out spread
   int transport = 0;              // MI355RT_TRANSPORT=peer/rccl: the exchange step of a multi-GPU frame (0: RCCL where it applies)
   int bounce_sync = 0;            // MI355RT_BOUNCE_SYNC=1: read the bounce queue's size back after every bounce (round 2's loop)
   int device_group = -1;          // MI355RT_GPU_GROUP=0/1: leaf grouping on the host / on the GPU (-1: GPU where it applies)
};

Switches switches();   // a copy of the current values (parsed from the environment on first use)

// name with or without the MI355RT_ prefix; value == nullptr: the switch's built-in default; name == nullptr: every switch
// back to what the environment said when the library first looked.  Returns false for a name the library does not know.
bool set_switch(const char *name, const char *value);

}   // namespace mi355rt
