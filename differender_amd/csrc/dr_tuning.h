// dr_tuning.h -- workgroup shapes, samples per lane, launch grids and pre-pass groups of the brick-centric kernels: the
// TUNING parameters. Every default is the measured optimum on MI355X (same-device A/B rows in profiles/r0N_ab_experiments.txt,
// quoted next to each); tools/mkvariant.sh overrides one with -DNAME=value to re-measure it. None of them changes a result.
#pragma once

// ---- threads per workgroup (FNT), ray segments listed per round (FEC <= FNT: one candidate per thread), waves per SIMD the
// ---- registers must allow (launch bounds) -- gathered in FlatCfg (march_flat.hip)
// forward: 30 KB of LDS, 96 VGPRs -> five four-wave workgroups per CU
#ifndef DR_FNT_FWD
#define DR_FNT_FWD 256
#endif
#ifndef DR_FEC_FWD
#define DR_FEC_FWD 256
#endif
#ifndef DR_FWD_WAVES
#define DR_FWD_WAVES 5
#endif
// backward with a gradient box (B1): 67 KB of LDS, 128 VGPRs -> two 8-wave workgroups per CU
#ifndef DR_FNT_BWD
#define DR_FNT_BWD 512
#endif
#ifndef DR_FEC_BWD
#define DR_FEC_BWD 256
#endif
#ifndef DR_BWD_WAVES
#define DR_BWD_WAVES 4
#endif
#ifndef DR_BWD_UNEVEN
#define DR_BWD_UNEVEN 7   // candidates dealt to a later wave for every 8 of an earlier one (0: even); 7: -1.1 %, 6: -0.5 %, 5: +2 %
#endif
// brick-centric backward w.r.t. the TF only (no gradient box, 33 KB of LDS): FOUR-wave workgroups, four per CU -- 3.22 ms against
// 3.80 with the 8-wave shape at 512^3 (profiles/r03_ab_experiments.txt); 96 VGPRs for a fifth workgroup spill 132 B per lane: 4.08 ms
#ifndef DR_FNT_BWDTF
#define DR_FNT_BWDTF 256
#endif
#ifndef DR_FEC_BWDTF
#define DR_FEC_BWDTF 128
#endif
#ifndef DR_BWDTF_WAVES
#define DR_BWDTF_WAVES 4
#endif
// alpha pre-pass at sampling rates >= 3: 192-entry tables + alpha-only TF table + 80 VGPRs = SIX workgroups per CU (at rate 8 a
// segment holds ~400 samples and occupancy is what the short, latency-bound workgroups lack: demo loop 10.35 -> 10.14 ms; at rate 1
// a brick's ~196 candidates would take two listing rounds: 512^3 tf1 forward +2.5 %; profiles/r05_ab_experiments.txt)
#ifndef DR_FEC_ALPHA_HI
#define DR_FEC_ALPHA_HI 192   // (25.3 KB at R = 256; 160 and 208 entries within 0.4 %)
#endif
#ifndef DR_ALPHA_WAVES_HI
#define DR_ALPHA_WAVES_HI 6
#endif

// ---- consecutive samples per lane: the cross-lane scan and the chunk bookkeeping are paid once per K * 64 samples, but lanes K
// ---- samples apart share fewer LDS words. Measured at 512^3: K = 2 wins at rate 1 (-3 %), K = 4 from 2 on (-13 .. -16 %); 8 loses
#ifndef DR_FWD_K
#define DR_FWD_K 2      // forward, sampling rates below 1.75
#endif
#ifndef DR_FWD_K_HI
#define DR_FWD_K_HI 4   // ... from 1.75 on
#endif
#ifndef DR_ALPHA_K
#define DR_ALPHA_K 4    // alpha pre-pass
#endif
#ifndef DR_BWDTF_K
#define DR_BWDTF_K 2    // brick-centric TF-only backward (what must survive the scan is six registers per sample)
#endif

// ---- alpha pre-pass: groups of brick layers, front to back (later groups skip terminated rays)
#ifndef DR_PP_GROUPS
#define DR_PP_GROUPS 6      // sampling rates >= 3 (3, 12 and 24 groups are all slower: profiles/r02_ab_experiments.txt)
#endif
#ifndef DR_PP_GROUPS_LO
#define DR_PP_GROUPS_LO 3   // below 3, under DR_HINT_EARLY_TERMINATION (2 / 3 / 4 / 6 / 9 groups at rate 1: demo forward 1.75 / 1.63 /
                            // 1.63 / 1.68 / 1.80 ms, 512^3 tf1 forward 1.54 / 1.42 / 1.43 / 1.45 / 1.51, CT-like 1.35 / 1.40 / 1.44 / 1.50 / 1.62)
#endif
#ifndef DR_UNLIT_SKIP
#define DR_UNLIT_SKIP 2     // the colour march drops segments the pre-pass found unlit: 0 never, 1 non-differentiable renders, 2 all
#endif

// ---- overflow work items of heavy bricks: workgroups of the second launch (what is resident at once on 256 CUs), items per ticket
#ifndef DR_ITEM_GRID_FWD
#define DR_ITEM_GRID_FWD 1280
#endif
#ifndef DR_ITEM_GRID_BWD
#define DR_ITEM_GRID_BWD 512
#endif
#ifndef DR_ITEM_RUN
#define DR_ITEM_RUN 2   // camera inside a 512^3 volume, fwd / bwd ms: static striding 11.3 / 19.8; runs of 1: 10.0 / 17.9, 2: 9.2 / 16.2, 3: 9.3 / 16.5, 4: 9.9 / 17.4
#endif

// ---- per-ray passes (ray_passes.hip, tf_tape.hip)
#ifndef DR_F2_WIDE
#define DR_F2_WIDE 4            // layers per step of F2's walk (their loads issued together)
#endif
#ifndef DR_CROSS_BATCH
#define DR_CROSS_BATCH 4        // 64-sample passes of the exact walk whose gathers are in flight together (8 and 16: no faster)
#endif
#ifndef DR_CROSS_GRID
#define DR_CROSS_GRID 8192      // workgroups of the crossing search over all views (2 048: +0.1 ms on the demo loop)
#endif
#ifndef DR_CROSS_QUAD_BELOW
#define DR_CROSS_QUAD_BELOW 2.0f   // sampling rate below which the crossing search runs four rays per wave
#endif
#ifndef DR_EXACT_GRID
#define DR_EXACT_GRID 1280      // four-wave workgroups of ray_exact_kernel when many rays are listed (five per CU)
#endif
#ifndef DR_EXACT_BWD_GRID
#define DR_EXACT_BWD_GRID 1024 // four-wave workgroups of ray_exact_bwd_kernel: four per CU (123 VGPRs)
#endif
#ifndef DR_TAPE_GRID
#define DR_TAPE_GRID 2560       // workgroups of tf_tape_bwd_kernel over all views (ten per CU: 12 KB of LDS at R = 256)
#endif
