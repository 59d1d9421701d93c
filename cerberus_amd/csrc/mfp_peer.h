// mfp_peer.h — device-side halo and dt transport between ranks (one process per GPU, or several processes on
// one GPU): every rank owns an *arena* of device memory that its neighbours map (HIP IPC across processes, a
// plain device pointer inside one process) and write into directly.  No host code runs per step:
//
//   halo exchange of dimension d, message number m (the same on every rank: the ranks run the same sequence):
//     sender    k_peer_pack    valid layers next to face (d, side) -> the neighbour's receive slot (d, 1 - side, m & 1)
//               k_peer_signal  system-scope release, then the neighbour's word ready[d][1 - side] = m
//     receiver  k_peer_wait    spins (bounded) until its own ready[d][side] >= m
//               k_peer_unpack  receive slot (d, side, m & 1) -> ghost layers
//   Two slots per face are enough without an acknowledgement: a rank writes message m + 2 into the slot of message m
//   only after it has waited for the neighbour's message m + 1, which the neighbour sent after it had unpacked m
//   (stream order on both sides).
//
//   dt minimum (MFP::estTimeStep's ParallelDescriptor::ReduceRealMin, MFP_timestep.cpp:18), message m:
//     k_peer_dt_put  this rank's dt -> slot [rank][m & 1] of every other rank, then their word dtready[rank] = m
//     k_peer_dt_min  waits for dtready[p] >= m of every p, dt = min over the slots and its own value
//
// Data and words cross the boundary with system-scope stores and loads (they bypass the non-coherent L2 of the
// XCDs), so nothing relies on what a kernel boundary flushes.  Every spin is bounded in wall-clock time
// (wall_clock64): a rank whose partner never arrives sets bit 1 of the context's error word and goes on, so that
// every wave of every launch ends and the failure is reported by the next synchronising call.
#pragma once
#include <hip/hip_runtime.h>

#include "mfp_types.h"

namespace cmfp_peer {
using cmfp::Geo;
using cmfp::IBox;

constexpr int MAX_RANKS = 64;
constexpr int ERR_TIMEOUT_BIT = 2;  // bit set in the context's error word when a wait gave up

// Byte offsets inside an arena; identical on every rank (same block shape, same states).
struct Layout {
    size_t ready;       // [3][2] unsigned long long
    size_t dtready;     // [MAX_RANKS] unsigned long long
    size_t dtslot;      // [MAX_RANKS][2] double
    size_t recv[3][2][2];  // [dim][side][parity]: doubles of every state, back to back
    size_t bytes;
};

struct Flags2 {
    int n;
    unsigned long long* flag[2];
    unsigned long long value;
};

struct DtPeers {
    int rank, world;
    char* arena[MAX_RANKS];  // arena[rank] = the local one
    size_t dtready, dtslot;
    unsigned long long value;
};

void pack(const Geo& g, int nc, int ngs, const double* A, int d, int side, double* remote_slot, hipStream_t st);
void unpack(const Geo& g, int nc, int ngs, double* A, int d, int side, const double* local_slot, hipStream_t st);
void signal(const Flags2& f, hipStream_t st);
void wait(const Flags2& f, int* err, double timeout_s, hipStream_t st);
void dt_put(const DtPeers& p, const double* dt, hipStream_t st);
void dt_min(const DtPeers& p, double* dt, int* err, double timeout_s, hipStream_t st);

}  // namespace cmfp_peer
