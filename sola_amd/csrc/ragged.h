// Ragged batches (samples of different N tracks x T frames x L text tokens in one pass): host-side shape bookkeeping and the
// device tables every shape-dependent kernel reads.  Shared by the ragged inference forward (forward_ragged.hip) and the
// ragged training step (forward.hip / backward.hip with a RagTables argument).
//
// The reference handles one sample per call (configs/mevis/default.yaml:37,42,47 batch_size 1; train.py:62-137,
// inference.py:44-58); here the token rows of all samples are concatenated and what depends on a sample's extent is
// described by small tables built on the device (ragged_plan_kernel) from the uploaded (N, T, L) arrays.
#pragma once
#include <vector>

#include "ctx.h"

struct RagShape {
    int V = 0, S = 0;
    std::vector<int> vN, vT[7], vRow0[7], vTrk0, vTp0;
    std::vector<int> sVid, sL, sLin0, sLrow0, sTrk0, sRow0, sTp0;
    long long rows[7] = {0};  // token rows per encoder level over the videos (level 0 = the object tokens)
    int NT = 0, maxN = 0, maxT[7] = {0}, maxW = 0;
    long long Mv = 0, Ms = 0, LW = 0, Lin = 0;
    int maxRowsSample = 0, sumTpV = 0, sumTpS = 0, sumNS = 0;
    bool identity = false;  // one sample per video, in order: the per-sample rows ARE the per-video rows
};
int rag_shape(const SolaCtx* c, const SolaRaggedBatch* b, RagShape& r);

// bytes of the table region (descriptor blob + every table); train adds the imap tables
size_t rag_tables_bytes(const RagShape& r, bool train);
// Lays the tables out at `base` (256-byte aligned, rag_tables_bytes long), uploads the descriptors through the context's pinned
// staging ring and launches the plan kernel on `s`; fills `out`.
int rag_build_tables(SolaCtx* c, const RagShape& r, char* base, bool train, RagTables* out, hipStream_t s);
// scratch of the sliced GroupNorm shape for a ragged batch (8 bytes per (unit, slice))
size_t rag_gn_slots_bytes(const RagShape& r);
