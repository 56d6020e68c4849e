// Device-resident stream front end: the part of libvorbis that sits between the application's PCM
// and vorbis_analysis() — vorbis_analysis_buffer / _wrote / _blockout (reference lib/block.c:405-812)
// and the envelope detector that drives block switching (lib/envelope.c) — for S streams at once.
//
// Every stream keeps the reference's own coordinates (centerW, pcm_current, ve_current, ... are the
// same numbers the reference would hold), so the decisions can be checked value by value.  The
// PCM of a channel lives in one of two buffers of `cap` floats; where the reference memmove()s the
// buffer down after a block (lib/block.c:757-759), the stream's origin `base` inside the buffer moves up
// instead, and only when it has passed base_max does the shift kernel copy the live samples to the start of
// the other buffer and flip the stream's parity (one copy per several blocks instead of one per block).
#pragma once
#include <stdint.h>
#include "setup.h"
#include "vorbis_mi355x.h"   /* vbm_packet_info */

#define VBM_FE_CHUNK 32          /* search steps evaluated per launch of the envelope kernels */

struct vbm_fe_state {
    int S, ch;
    long cap;                    // floats per channel buffer
    long plane;                  // S*ch*cap
    int marks;                   // entries of the per-stream mark ring (cap/64 + 8)
    float *pcm;                  // [2][S*ch][cap]
    int *parity;                 // [S]
    int *base;                   // [S] where the stream's sample 0 sits in its channel buffers (see k_fe_shift)
    int base_max;                // the buffers are compacted once base would pass this
    int *overflow;               // [1] writes refused because a stream's buffer was full (device-built rounds: no host check)
    // vorbis_dsp_state (lib/block.c:173-344 initial values)
    int *pcm_current, *centerW, *lW, *W, *nW, *eofflag, *preextrapolate;   // [S]
    long long *granulepos, *sequence;                                      // [S]
    // envelope_lookup (lib/envelope.h:56-76); SoA with the stream / channel index innermost
    int *ve_current, *ve_cursor, *ve_curmark, *ve_stretch;                 // [S]
    int *ve_mark;                // [marks][S]
    float *ve_ampbuf;            // [VE_AMP][S*ch][16]  band innermost (12 used): the 16 lanes of a stream read one run
    int *ve_ampptr;              // [S*ch][16]
    float *ve_nearDC;            // [VE_NEARDC][S*ch]
    float *ve_nearacc;           // [2][S*ch]   nearDC_acc, nearDC_partialacc
    int *ve_nearptr;             // [S*ch]
    int *ve_first, *ve_last;     // [S] search steps [first, last) still to evaluate
    float *ve_spec;              // [S*ch][VBM_FE_CHUNK][64] spectra of the 128-point search MDCTs
};

// one per stream and round (host readable)
struct vbm_fe_decision {
    int ready;                   // vorbis_analysis_blockout returned 1
    int lW, W, nW;
    int block_mode;              // blocktype | W << 1   (lib/mapping0.c:768-775)
    int eos;                     // vb->eofflag
    int beginW;                  // first sample of the block in the (pre-shift) buffer
    int movement;                // samples the buffer moves down after this block
    long long granulepos, sequence;
};

// One round built on the device (k_fe_classify / k_fe_plan / k_fe_commit): block type m owns the lane region
// [lane0[m], lane0[m] + cap[m]) of the round's lists and of the encoder workspace behind them.
struct vbm_fe_round {
    int lane0[4], cap[4];
    int first_round;             // first round of a call (its long blocks form the call's big batch)
    int *count;                  // [4] device: blocks of each type this round (<= cap)
    int *slot;                   // [S] lane of the stream's block this round, -1 none
    int *stream_id;              // [lanes]  (the encoder workspace's own list)
    uint8_t *wflags;             // [lanes]
    int *begin;                  // [lanes] first sample of the block in its stream's buffer
    vbm_packet_info *info;       // [lanes] description of every lane's block (stream = -1: none)
    unsigned long long *stats;   // [5] running totals: blocks of type 0..3, samples the streams advanced by
};

