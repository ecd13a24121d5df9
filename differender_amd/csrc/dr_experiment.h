// dr_experiment.h -- every build switch that changes what the kernels COMPUTE (timing-only what-ifs with wrong results) or
// instruments them (diagnostic clocks / counters in the workspace header) is declared here, and a translation unit compiled with
// any of them marks the whole library: dr_build_flags() reports it, dr_abi_version() turns NEGATIVE, and the Python loader
// (differender_amd/_native.py) refuses the library unless DIFFERENDER_ALLOW_EXPERIMENT=1 (an ablation build
// loaded through DIFFERENDER_HIP_LIB used to pass the version check and return wrong gradients silently).
// Included by every translation unit (through dr_device.h / dr_kernels.h). The shipped library defines none of the switches.
#pragma once

// ---- what-ifs with WRONG results (tools/README.md; findings in profiles/r0N_ab_experiments.txt) ----
#if defined(DR_ABL_NOSCATTER) || defined(DR_ABL_NOATOMIC) || defined(DR_ABL_NOFLUSH) || defined(DR_ABL_NOBARRIER) || \
    defined(DR_ABL_HALFREADS) || defined(DR_ABL_XREUSE) || defined(DR_ABL_SMALLBOX) || defined(DR_ABL_ALPHA13) || defined(DR_ABL_NOMEMSET) || defined(DR_ABL_NOITEMS) || \
    defined(DR_CROSS_NORESTART) || defined(DR_ABL_WRONG) || defined(DR_D4_DEBUG) || defined(DR_D4_BUDGET_OVERRIDE)
#define DR_EXPERIMENT_WRONG 1
#else
#define DR_EXPERIMENT_WRONG 0
#endif
// ---- right results, but not the shipped kernels: LDS ballast, instrumentation, launch shapes measured slower ----
#if (defined(DR_ABL_EXTRA_LDS_BWD) && DR_ABL_EXTRA_LDS_BWD != 0) || (defined(DR_ABL_EXTRA_LDS_ALPHA) && DR_ABL_EXTRA_LDS_ALPHA != 0) || \
    (defined(DR_PHASE_TIMING) && DR_PHASE_TIMING != 0) || defined(DR_LANE_STATS) || defined(DR_CROSS_STATS) || defined(DR_VIEW_FASTEST) || (defined(DR_F1_RESIDENT) && DR_F1_RESIDENT != 0)
#define DR_EXPERIMENT_DIAG 1
#else
#define DR_EXPERIMENT_DIAG 0
#endif

// DR_BUILD_FLAGS_DROPPED: the Makefile's probe found that this compiler does not know one of the two tuned -mllvm flags and built
// without it (right results, slower kernels): bit 2 of dr_build_flags(), so that a bench line can say which build it timed
#ifndef DR_BUILD_FLAGS_DROPPED
#define DR_BUILD_FLAGS_DROPPED 0
#endif
enum { DR_BUILD_WRONG_RESULTS = 1, DR_BUILD_DIAGNOSTIC = 2, DR_BUILD_UNTUNED = 4 };

extern "C" int dr_experiment_flags_;   // defined in capi.hip; OR-ed at load time by the marked translation units
#if DR_EXPERIMENT_WRONG || DR_EXPERIMENT_DIAG || DR_BUILD_FLAGS_DROPPED
namespace {
struct DrExperimentMark {
    DrExperimentMark() { dr_experiment_flags_ |= (DR_EXPERIMENT_WRONG ? DR_BUILD_WRONG_RESULTS : 0) | (DR_EXPERIMENT_DIAG ? DR_BUILD_DIAGNOSTIC : 0) | (DR_BUILD_FLAGS_DROPPED ? DR_BUILD_UNTUNED : 0); }
};
static DrExperimentMark dr_experiment_mark_;   // one per marked translation unit, runs when the library is loaded
}  // namespace
#endif
