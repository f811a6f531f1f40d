// betaone_amd/csrc/bo_replay.h -- GPU-resident replay buffer: finished games as compact records in HBM, training batches expanded on
// the device.
//
// The reference hands self-play results to training through pickles of DENSE tuples (49.4 KB per ply, self_play.py:220-231), reads
// them all back into one Python list (train.load_recent_data, train.py:187-219) and lets a DataLoader index it
// (ChessDataset.__getitem__, train.py:179-184; the loop that consumes the batches: train_network, train.py:252-262).  Here a ply is
// its position (80 B), its repetition count under the END-of-game tracker (self_play.py:200-208: the training planes of ply i are
// encoded with the tracker of the finished game), its sparse pi and its z: ~110 B per ply resident in HBM (100 M plies = 11 GB of the
// 288), and a batch is made where it is consumed -- one wave per sampled ply writes the 120 planes, the dense pi row and z.
//   bo_k_replay_counts   once per added game: repetition count of every position of the game (a position's count is the number of
//                        key-equal positions in the WHOLE game minus one: what bo_k_encode_positions recomputes per encoded ply)
//   bo_k_replay_encode   one wave per sample: 8 history blocks from the ply's own game (positions are stored game-contiguous),
//                        scalar planes, dense pi row (zero fill + scatter), z
#pragma once
#include "bo_tree.h"

BO_KERNEL void bo_k_replay_counts(const DPos *pos, int n_pos, int *rep) {
    const int i = bo_block(), s = bo_lane();
    const DPos H = pos[i];
    int c = 0;
    for (int j = s; j < n_pos; j += 64) c += key_equal(pos[j], H) ? 1 : 0;
    c = bo_wave_sum(c);
    if (s == 0) rep[i] = c > 1 ? c - 1 : 0;
}

// sample i = the record in ring slot s_slot[i], the s_k[i]-th ply of its game (so slots s_slot - min(7, s_k) .. s_slot are its history)
BO_KERNEL void bo_k_replay_encode(const DPos *pos, const int *rep, const int *pi_n, const int *pi_idx, const float *pi_val, const float *z,
                                  int W, const int *s_slot, const int *s_k, float *states, float *pis, float *zs) {
    const int b = bo_block(), s = bo_lane();
    const int slot = s_slot[b], k = s_k[b];
    float *row = states + (size_t)b * BO_ROW;
    const int nb = k < 7 ? k + 1 : 8, h0 = slot - (nb - 1);
    for (int pl = 0; pl < (8 - nb) * 14; pl++) row[pl * 64 + s] = 0.0f;
    for (int j = 0; j < nb; j++) encode_block(row, 8 - nb + j, pos[h0 + j], rep[h0 + j]);
    encode_scalars(row, pos[slot]);
    // dense pi row: every address has ONE writer (lane = action mod 64), which looks its action up among the ply's few entries
    float *pr = pis + (size_t)b * BO_NUM_ACTIONS;
    const int n = pi_n[slot];
    const int *ix = pi_idx + (size_t)slot * W;
    const float *vx = pi_val + (size_t)slot * W;
    for (int a = s; a < BO_NUM_ACTIONS; a += 64) {
        float v = 0.0f;
        for (int e = 0; e < n; e++) v = ix[e] == a ? vx[e] : v;
        pr[a] = v;
    }
    if (s == 0) zs[b] = z[slot];
}
