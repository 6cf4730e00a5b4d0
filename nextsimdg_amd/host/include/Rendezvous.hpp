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
//! ranks).  Rank 0's own buffer is not passed to the sink.  The server listens on port + 18.
//! ACKNOWLEDGED: once every rank has delivered, rank 0 runs `commit` (the write of the restart file) while the
//! connections are still open and then answers every sender with one status byte.  A sender returns only after that
//! byte: if rank 0 failed -- the sink or `commit` threw, a rank reported a failure, a rank never connected -- every
//! sender throws std::runtime_error too, so that no rank of a run without a restart file exits with status 0.
//! A rank that has died never connects: rank 0 then throws after `timeoutSeconds` (and writes nothing), and a sender
//! whose rank 0 has gone throws in the same way.  `deadlineSeconds` <= 0 means the communicator's "wait for ever"
//! (NSDG_COMM_TIMEOUT_S = 0): a day.
void gatherToRankZero(const RankEnvironment& env, const void* buffer, std::size_t bytes,
    const std::function<void(int rank, const char* data, std::size_t count)>& sink, int timeoutSeconds = 120,
    const std::function<void()>& commit = nullptr);

//! What a rank r > 0 calls INSTEAD of gatherToRankZero when it cannot deliver (its own run failed): rank 0 learns of it at
//! once -- its gatherToRankZero throws without waiting for the timeout -- and tells the other ranks.  Never throws.
void reportFailureToRankZero(const RankEnvironment& env, int timeoutSeconds = 10) noexcept;

} // namespace Nextsim
