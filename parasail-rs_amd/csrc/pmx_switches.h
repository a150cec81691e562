// pmx_switches.h -- every environment switch of the library, in one table.
//
// The dispatcher picks one implementation per call from the shapes and the scoring; a switch overrides one of those choices.
// All alternatives produce identical results: tests/test_gpu_switches.py runs a fixed set of batches under every "force" switch
// listed here (it reads the table through pmx_switches()) and compares the records with the unswitched run, and the A/B scripts under
// profiles/ use them to time one alternative against another.  Nothing else in the library reads the environment.
// The environment is read at every call (tests flip switches between calls); a lookup costs ~50 ns.
#pragma once
#include <cstdlib>

struct PmxSwitchDoc { const char *name, *kind, *what; };

// kind: "force" = boolean, forces an alternative implementation; "value" = numeric parameter; "path"; "diag" = diagnostics output;
//       "convention" = boolean, changes an UNPINNED output convention (not result-neutral: include/pmx_conventions.h)
#define PMX_SWITCH_TABLE(X) \
    X("PMX_MATRIX_DIR",               "path",  "directory of NCBI-format matrix files for names that are not built in (parasail_matrix_lookup)") \
    X("PMX_SW16_VARIANT",             "value", "local kernels: highest arithmetic variant allowed (0 saturating int16, 1 max3, 2 max3 + 32-bit add/sub)") \
    X("PMX_SW16_NO_U8",               "force", "local kernels: int16 profile entries instead of bytes") \
    X("PMX_SW16_NO_SKEW",             "force", "local kernels: no column skew (E extension with a subtract)") \
    X("PMX_SW16_NO_PERMTABLE",        "force", "local kernels, alphabets of <= 4 letters: LDS profile instead of the v_perm score table") \
    X("PMX_SW16_NO_SHARED",           "force", "local, profile arm: per-pair profiles instead of the workgroup-shared profile kernel") \
    X("PMX_SW16_NO_MATRIX_LOOKUP",    "force", "local, large alphabets: LDS profiles instead of the matrix-lookup kernel") \
    X("PMX_SW16_MATRIX_LOOKUP",       "force", "local: matrix-lookup kernel also for small alphabets") \
    X("PMX_SW16_NO_FETCH",            "force", "local, references >= 1024: stage references in LDS instead of fetching them two steps ahead") \
    X("PMX_NO_FAST_NWSG",             "force", "global / semi-global: general int32 kernel instead of the packed int16 kernels") \
    X("PMX_NWSG8_GENERAL",            "force", "global / semi-global at width 8: general int32 kernel instead of the packed int16 kernel with range tracking") \
    X("PMX_NWSG16_NO_PERMTABLE",      "force", "global / semi-global, alphabets of <= 4 letters: LDS profiles instead of the v_perm score table (top-aligned form)") \
    X("PMX_NWSG16_GEN1",              "force", "global / semi-global: first-generation packed kernel") \
    X("PMX_NWSG16_NO_SHARED",         "force", "global / semi-global, profile arm: per-pair profiles instead of the shared-profile kernel") \
    X("PMX_NWSG16_NO_MATRIX_LOOKUP",  "force", "global / semi-global, large alphabets: LDS profiles instead of the matrix-lookup kernel") \
    X("PMX_NWSG16_NO_FETCH",          "force", "global / semi-global, references >= 1024: stage references in LDS") \
    X("PMX_NO_FAST_STATS",            "force", "statistics: general kernel instead of the packed statistics kernels") \
    X("PMX_STATS16_GEN1",             "force", "statistics: first-generation statistics kernel instead of stats16p") \
    X("PMX_STATS16P_ALWAYS",          "force", "statistics: stats16p also where statistics by traceback would be chosen") \
    X("PMX_STATS16P_NO_MATRIX_LOOKUP","force", "statistics, large alphabets: LDS profiles instead of the matrix lookup") \
    X("PMX_STATS_BY_TRACE",           "force", "statistics by traceback also for small batches") \
    X("PMX_STATS_BY_TRACE_ANY",       "force", "statistics by traceback also for large alphabets with long references") \
    X("PMX_NO_STATS_BY_TRACE",        "force", "statistics: never by traceback (statistics-carrying kernels)") \
    X("PMX_NWSGQ_ENDS_ALWAYS",        "force", "shared-profile traceback sweep <16,20>, global alignment: the instance with the free-end captures compiled in (two waves per SIMD)") \
    X("PMX_NWSGQ_NO_R19",             "force", "statistics by traceback, profile arm: the <16,20> shape instead of <16,19> for queries of 256-303 rows") \
    X("PMX_NWSGQ_NO_R20",             "force", "statistics by traceback, profile arm: the <32,10> shape instead of <16,20> for queries of 256-319 rows") \
    X("PMX_STATS_CHUNK_BYTES",        "value", "statistics by traceback: bytes of trace scratch per chunk (tests force several chunks)") \
    X("PMX_STATS_EQUAL_CHUNKS",       "force", "statistics by traceback, shared profile: chunks of equal size instead of whole rounds of resident workgroups") \
    X("PMX_STATS_TAIL_LAST",          "force", "statistics by traceback, profile arm: the remainder after the whole rounds runs last on the shared trace buffers instead of first on a buffer of its own") \
    X("PMX_STATS_NO_SHORT_TAIL",      "force", "statistics by traceback, shared profile: the remainder chunk on the same shape as the whole rounds (not the half-length waves of <32,10>)") \
    X("PMX_STATS_NO_OVERLAP",         "force", "statistics by traceback: sweep and walk of every chunk back to back on one stream (no double buffering)") \
    X("PMX_NO_FAST_TRACE",            "force", "traceback: general kernel (1 byte per cell) instead of the packed 4-bit kernels") \
    X("PMX_TRACE16_GEN1",             "force", "traceback: first-generation packed kernel and one-lane walk") \
    X("PMX_TRACE_NO_BFI",             "force", "traceback: three-instruction decision merge instead of the bounded-difference v_bfi merge") \
    X("PMX_TRACE_FETCH",              "force", "traceback <16,16>: fetch references from HBM instead of staging them") \
    X("PMX_CIGAR_CHUNK_BYTES",        "value", "batch CIGAR: bytes of trace scratch per chunk (tests force several chunks)") \
    X("PMX_CIGAR_NO_OVERLAP",         "force", "batch CIGAR: sweep and walk back to back on one stream (no double buffering)") \
    X("PMX_NO_FAST_BANDED",           "force", "banded: general kernel with a band mask instead of the band-only kernel") \
    X("PMX_BANDED_NO_STRIP",          "force", "banded, alphabets of <= 4 letters: the anti-diagonal kernels of pmx_banded.hip instead of the band-strip kernel (C offsets per lane)") \
    X("PMX_BSTRIP_SHAPE",             "value", "band-strip kernel: lanes per pair x offsets per lane, e.g. 4x8 (ignored unless that shape exists and holds the band)") \
    X("PMX_BSTRIP_CELL_GUARDS",       "force", "band-strip kernel <8,13>, band 48: a guard per cell instead of the guarded block in front of the band") \
    X("PMX_BSTRIP_TIES_INLINE",       "force", "band-strip kernel, local alignment: ties that could move an end cell are settled where they occur (one launch) instead of remembered and redone by a second launch") \
    X("PMX_BSTRIP_ONE_SKEW",          "force", "band-strip kernel, global / semi-global: one skew (F pays a subtraction per cell) instead of the double skew") \
    X("PMX_BANDED_NO_PACKED",         "force", "banded local alignment: the 32-bit staged kernel instead of the packed int16 kernel (two pairs per lane group)") \
    X("PMX_BANDED_NO_ROWPERM",        "force", "packed banded kernel, alphabets of <= 7 letters: one LDS byte lookup per cell instead of the 8-byte matrix row + v_perm") \
    X("PMX_BANDED_NO_SHARED_ROWS",    "force", "packed banded kernel, one shared query over <= 7 letters: each pair on its own query rows (forms 1 / 0) instead of both pairs of a lane group on the same rows") \
    X("PMX_BANDED_NO_STAGING",        "force", "banded: per-cell kernel (symbols from HBM) instead of the LDS-staged kernel with the lean interior loop") \
    X("PMX_GENERAL_ONE_WAVE",         "force", "general kernel: one wave per pair also for few long pairs (no pipelined sharing of a pair among the waves of a workgroup)") \
    X("PMX_GENERAL_CHUNK_BYTES",      "value", "batches through the general kernel (score fallback, score tables): bytes of boundary scratch per chunk (tests force one-pair chunks)") \
    X("PMX_NO_LONG_KERNEL",           "force", "few long pairs: the per-pair kernels instead of the kernel that spreads one pair's query bands over the chip") \
    X("PMX_LONG_ROWS_PER_LANE",       "value", "long-pair kernel: rows per lane (2, 4 or 16: bands of 128, 256 or 1 024 query rows)") \
    X("PMX_LONG_TWO_COLUMNS",         "force", "long-pair kernel: two columns per step also where the dispatcher's time model prefers one (batches, square pairs)") \
    X("PMX_LONG_ONE_COLUMN",          "force", "long-pair kernel: one column per step (the first form) instead of two") \
    X("PMX_LONG_CHUNK_COLS",          "value", "long-pair kernel: boundary columns a band takes over from the band above at a time (16 or 64)") \
    X("PMX_LONG_MIN_CELLS",           "value", "cells of the largest pair from which a call of at most 16 pairs (queries of 512 rows or more) takes the long-pair kernel") \
    X("PMX_LONG_SPIN_LIMIT",          "value", "long-pair kernel: polls a band waits for the band above before the launch gives up and the call is redone on the per-pair kernels (0: any wait gives up)") \
    X("PMX_LONG_CHUNK_BYTES",         "value", "long-pair kernel over a batch: bytes of boundary scratch per chunk (tests force several chunks)") \
    X("PMX_NO_FAST_TABLE",            "force", "score tables: general kernel instead of the table kernel") \
    X("PMX_CIGAR_SWAP_ID",            "convention", "CIGAR letters / BAM ops of the two gap states exchanged (I <-> D) in get_cigar, ssw and batch CIGAR text") \
    X("PMX_DEFER_ALIGN",              "force", "one-pair alignment functions (score / statistics names): the call queues the pair and returns a pending result; the first accessor of any of the thread's pending results runs the queue as one batch") \
    X("PMX_TIMING",                   "diag",  "stage times of the batch CIGAR host entry on stderr")

const char *pmx_env(const char *name);        // the environment value of a registered switch (pmx_api.hip; an unregistered name aborts)
