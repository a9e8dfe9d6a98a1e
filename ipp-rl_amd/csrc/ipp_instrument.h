// Instrumentation hooks of the kernels -- every one a no-op in the product build.  The measurement builds of tools/ and of
// `make timeline` define the switches below; the kernels themselves only carry the one-line macro calls.
//   IPP_PHASE_TIMING   band-tile kernels: per-phase wall-clock sums (ipp_streamed_bytes prints them)
//   IPP_TIMELINE       patch kernels: per-item marks and per-unit trace of the last launch (tools/timeline*.py)
//   IPP_WAVE_CLOCKS    (on top of IPP_TIMELINE) per-wave phase clocks of the unit loop (tools/wave_phases.py)
//   IPP_EXIT_POINTS    instruction / traffic counts by section: the launch returns at exit point dbg_capture - 1 or skips section
//                      dbg_capture - 1 of the unit loop (tools/valu_sections.py); results are wrong from there on
// (the A/B variants of rounds 3-4 -- software-pipelined groups, stage-ordered bookkeeping, units over the patch stride, the
// ablation masks -- are gone from the source: profiles/r04_experiments.txt keeps their numbers, commit 245be41 their code)
#pragma once
// Debug builds only (-DIPP_PHASE_TIMING=1): thread 0 of every workgroup adds the time since the previous tick
// (100 MHz wall clock) to counters[k]; ipp_streamed_bytes prints the sums.
#ifndef IPP_PHASE_TIMING
#define IPP_PHASE_TIMING 0
#endif
#if IPP_PHASE_TIMING
#define IPP_TICK_DECL(t) unsigned long long t = wall_clock64()
#define IPP_TICK(v, k, t) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&(v).counters[k], n_ - (t)); (t) = n_; } } while (0)
#else
#define IPP_TICK_DECL(t) ((void)0)
#define IPP_TICK(v, k, t) ((void)0)
#endif
// -DIPP_TIMELINE=1: timeline of the last fused step launch: per item the wall clock at workgroup start, at the end
// of phase A and at the last wave's exit (three stores per workgroup; IPP_TIMELINE_FILE=<path> makes
// ipp_streamed_bytes dump it, tools/timeline.py prints the occupancy over time).
#ifndef IPP_TIMELINE
#define IPP_TIMELINE 0
#endif
#if IPP_TIMELINE
constexpr int kTimelineItems = 65536;
__device__ unsigned long long g_timeline[8 * kTimelineItems];  // per item: start, end of phase A, end, header done, observation done, gather done
#define IPP_MARK(item, k) do { if ((item) < kTimelineItems) g_timeline[8 * (item) + (k)] = wall_clock64(); } while (0)
// per-wave phase clocks of k_step_patch (shader clock, s_memtime; -DIPP_WAVE_CLOCKS=1 on top of IPP_TIMELINE: the clock reads
// serialise the wave and make the kernel several times slower -- proportions only): a wave sums the time between its ticks per
// phase in scalar registers and adds the sums to g_wphase when it exits; ipp_streamed_bytes prints and clears them
__device__ unsigned long long g_wphase[16];
// per-unit trace of the patch kernels: [item][wave 0..1][slot 0..7][unit index << 32 | rows, start, stream done, end] (wall clock, 10 ns)
constexpr int kUnitTraceItems = 4096;
__device__ unsigned long long g_unit_trace[kUnitTraceItems * 2 * 8 * 4];
#define IPP_UNIT_TRACE(item, wave, slot, k, val) do { if ((item) < kUnitTraceItems && (wave) < 2 && (slot) < 8 && (threadIdx.x & 63) == 0) \
    g_unit_trace[((((size_t)(item) * 2 + (wave)) * 8 + (slot)) * 4) + (k)] = (val); } while (0)
#ifndef IPP_WAVE_CLOCKS
#define IPP_WAVE_CLOCKS 0
#endif
#endif
#if IPP_TIMELINE && IPP_WAVE_CLOCKS
#define IPP_WT_DECL unsigned long long wt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wt_last_ = clock64()
// (IPP_WAVE_CLOCKS is a bit mask of the ticks to keep: interval k = time since the previous KEPT tick; two ticks per unit keep the
// perturbation small -- e.g. 0x1c6: stream = interval 2)
#define IPP_WT(k) do { if ((IPP_WAVE_CLOCKS >> (k)) & 1) { const unsigned long long n_ = clock64(); wt_[k] += n_ - wt_last_; wt_last_ = n_; } } while (0)
#define IPP_WT_COUNT(k, n) do { wt_[k] += (n); } while (0)
#define IPP_WT_RESET do { wt_last_ = clock64(); } while (0)
#define IPP_WT_FLUSH(lane) do { if ((lane) == 0) for (int q_ = 0; q_ < 12; ++q_) if (wt_[q_]) atomicAdd(&g_wphase[q_], wt_[q_]); } while (0)
#else
#if !IPP_TIMELINE
#define IPP_MARK(item, k) ((void)0)
#define IPP_UNIT_TRACE(item, wave, slot, k, val) ((void)0)
#endif
#define IPP_WT_DECL ((void)0)
#define IPP_WT(k) ((void)0)
#define IPP_WT_COUNT(k, n) ((void)0)
#define IPP_WT_RESET ((void)0)
#define IPP_WT_FLUSH(lane) ((void)0)
#endif

#ifndef IPP_EXIT_POINTS
#define IPP_EXIT_POINTS 0
#endif
#if IPP_EXIT_POINTS
#define IPP_EXIT_POINT(k) do { if (v.dbg_capture == (k) + 1) return; } while (0)
#define IPP_UNIT_SKIP(k) (v.dbg_capture == (k) + 1)
#else
#define IPP_EXIT_POINT(k) do { } while (0)
#define IPP_UNIT_SKIP(k) false
#endif
