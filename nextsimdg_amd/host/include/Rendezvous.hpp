// Rendezvous.hpp -- how the ranks of a multi-process run (one process per GPU) agree on the RCCL communicator
// id: rank 0 creates it (nsdg_comm_unique_id) and serves it over TCP to the other ranks.  The launcher provides
// the usual environment: WORLD_SIZE, RANK, LOCAL_RANK, MASTER_ADDR, MASTER_PORT (as set by torch.distributed.run,
// which also starts plain executables: `python -m torch.distributed.run --no-python --nproc-per-node N nextsim_amd ...`).
// The reference is a single process; this has no counterpart there.
#pragma once
#include <cstddef>
#include <functional>
#include <string>

namespace Nextsim {

struct RankEnvironment {
    int world = 1, rank = 0, localRank = 0;
    std::string masterAddr = "127.0.0.1";
    int masterPort = 29500;
    static RankEnvironment fromEnv(); //!< world == 1 when the variables are absent
};

//! Rank 0 sends `bytes` bytes of `buffer` to the world-1 other ranks, which receive them into `buffer`.
//! The server listens on port + 17 (the launcher's own store owns `port`).  Throws std::runtime_error
//! on any socket error or after `timeoutSeconds`.
void broadcastFromRankZero(const RankEnvironment& env, void* buffer, std::size_t bytes, int timeoutSeconds = 120);

//! The opposite direction, used once per run for the restart file: every rank r > 0 sends `bytes` bytes of `buffer`
//! to rank 0, which calls sink(r, data, count) for each of them (in the order they arrive; counts may differ between
//! ranks).  Rank 0's own buffer is not passed to the sink.  The server listens on port + 18.  A rank that has died
//! never connects: rank 0 then throws std::runtime_error after `timeoutSeconds` (no partial file is written by the
//! caller), and a sender whose rank 0 has gone throws in the same way.
void gatherToRankZero(const RankEnvironment& env, const void* buffer, std::size_t bytes,
    const std::function<void(int rank, const char* data, std::size_t count)>& sink, int timeoutSeconds = 120);

} // namespace Nextsim
