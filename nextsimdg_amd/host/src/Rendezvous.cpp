#include "Rendezvous.hpp"
#include <memory>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

namespace Nextsim {

namespace {
int envInt(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

void sendAll(int fd, const char* p, std::size_t n)
{
    while (n > 0) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0)
            throw std::runtime_error("rendezvous: send failed");
        p += k, n -= (std::size_t)k;
    }
}

void recvAll(int fd, char* p, std::size_t n)
{
    while (n > 0) {
        const ssize_t k = ::recv(fd, p, n, 0);
        if (k <= 0)
            throw std::runtime_error("rendezvous: connection closed or timed out before the data arrived");
        p += k, n -= (std::size_t)k;
    }
}

struct Socket {
    int fd = -1;
    ~Socket()
    {
        if (fd >= 0)
            ::close(fd);
    }
};
} // namespace

RankEnvironment RankEnvironment::fromEnv()
{
    RankEnvironment e;
    e.world = envInt("WORLD_SIZE", 1);
    e.rank = envInt("RANK", 0);
    e.localRank = envInt("LOCAL_RANK", e.rank);
    if (const char* a = std::getenv("MASTER_ADDR"))
        if (*a)
            e.masterAddr = a;
    e.masterPort = envInt("MASTER_PORT", 29500);
    if (e.world < 1 || e.rank < 0 || e.rank >= e.world)
        throw std::invalid_argument("RANK / WORLD_SIZE in the environment are inconsistent");
    return e;
}

void broadcastFromRankZero(const RankEnvironment& env, void* buffer, std::size_t bytes, int timeoutSeconds)
{
    if (env.world <= 1)
        return;
    const int port = env.masterPort + 17;
    sockaddr_in addr;
    std::memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET;
    addr.sin_port = htons((unsigned short)port);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeoutSeconds);
    if (env.rank == 0) {
        Socket srv;
        srv.fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (srv.fd < 0)
            throw std::runtime_error("rendezvous: socket() failed");
        const int one = 1;
        ::setsockopt(srv.fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        if (::bind(srv.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0 || ::listen(srv.fd, env.world) != 0)
            throw std::runtime_error("rendezvous: cannot listen on port " + std::to_string(port));
        timeval tv = { timeoutSeconds, 0 };
        ::setsockopt(srv.fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        for (int k = 1; k < env.world; ++k) {
            Socket c;
            c.fd = ::accept(srv.fd, nullptr, nullptr);
            if (c.fd < 0)
                throw std::runtime_error("rendezvous: only " + std::to_string(k - 1) + " of " + std::to_string(env.world - 1) + " ranks connected");
            sendAll(c.fd, static_cast<const char*>(buffer), bytes);
        }
        return;
    }
    if (::inet_pton(AF_INET, env.masterAddr.c_str(), &addr.sin_addr) != 1)
        throw std::runtime_error("rendezvous: MASTER_ADDR must be a dotted IPv4 address, got " + env.masterAddr);
    for (;;) { // rank 0 may not be listening yet
        Socket c;
        c.fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (c.fd < 0)
            throw std::runtime_error("rendezvous: socket() failed");
        if (::connect(c.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0) {
            timeval tv = { timeoutSeconds, 0 };
            ::setsockopt(c.fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            recvAll(c.fd, static_cast<char*>(buffer), bytes);
            return;
        }
        if (std::chrono::steady_clock::now() > deadline)
            throw std::runtime_error("rendezvous: rank 0 did not answer on " + env.masterAddr + ":" + std::to_string(port));
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
}

namespace {
struct GatherHeader {
    std::int64_t rank, bytes; // bytes < 0: the rank reports that it failed
};
int gatherSeconds(int timeoutSeconds) { return timeoutSeconds > 0 ? timeoutSeconds : 86400; }

// connects to rank 0's gather port (retrying until the deadline) and sends the header (+ payload)
bool connectAndSend(const RankEnvironment& env, const GatherHeader& h, const void* payload, int timeoutSeconds, Socket& c, std::string& why)
{
    const int port = env.masterPort + 18;
    sockaddr_in addr;
    std::memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET;
    addr.sin_port = htons((unsigned short)port);
    if (::inet_pton(AF_INET, env.masterAddr.c_str(), &addr.sin_addr) != 1) {
        why = "gather: MASTER_ADDR must be a dotted IPv4 address, got " + env.masterAddr;
        return false;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeoutSeconds);
    for (;;) { // rank 0 may not be listening yet
        c.fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (c.fd < 0) {
            why = "gather: socket() failed";
            return false;
        }
        if (::connect(c.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0) {
            timeval tv = { timeoutSeconds, 0 };
            ::setsockopt(c.fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
            ::setsockopt(c.fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            sendAll(c.fd, reinterpret_cast<const char*>(&h), sizeof h);
            if (h.bytes > 0)
                sendAll(c.fd, static_cast<const char*>(payload), (std::size_t)h.bytes);
            return true;
        }
        ::close(c.fd);
        c.fd = -1;
        if (std::chrono::steady_clock::now() > deadline) {
            why = "gather: rank 0 did not answer on " + env.masterAddr + ":" + std::to_string(port);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
}
} // namespace

void gatherToRankZero(const RankEnvironment& env, const void* buffer, std::size_t bytes,
    const std::function<void(int, const char*, std::size_t)>& sink, int timeoutSeconds, const std::function<void()>& commit)
{
    if (env.world <= 1) {
        if (commit)
            commit();
        return;
    }
    timeoutSeconds = gatherSeconds(timeoutSeconds);
    if (env.rank == 0) {
        const int port = env.masterPort + 18;
        sockaddr_in addr;
        std::memset(&addr, 0, sizeof addr);
        addr.sin_family = AF_INET;
        addr.sin_port = htons((unsigned short)port);
        Socket srv;
        srv.fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (srv.fd < 0)
            throw std::runtime_error("gather: socket() failed");
        const int one = 1;
        ::setsockopt(srv.fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        if (::bind(srv.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0 || ::listen(srv.fd, env.world) != 0)
            throw std::runtime_error("gather: cannot listen on port " + std::to_string(port));
        timeval tv = { timeoutSeconds, 0 };
        ::setsockopt(srv.fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        std::vector<bool> seen(env.world, false);
        std::vector<char> data;
        std::vector<std::unique_ptr<Socket>> open; // the senders wait for the status byte on these
        std::string failure;
        try {
            for (int k = 1; k < env.world; ++k) {
                std::unique_ptr<Socket> c(new Socket());
                c->fd = ::accept(srv.fd, nullptr, nullptr);
                if (c->fd < 0)
                    throw std::runtime_error("gather: only " + std::to_string(k - 1) + " of " + std::to_string(env.world - 1) + " ranks delivered their rows");
                ::setsockopt(c->fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
                ::setsockopt(c->fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
                GatherHeader h;
                recvAll(c->fd, reinterpret_cast<char*>(&h), sizeof h);
                open.push_back(std::move(c));
                if (h.rank < 1 || h.rank >= env.world || seen[(std::size_t)h.rank])
                    throw std::runtime_error("gather: unexpected sender (rank " + std::to_string(h.rank) + ")");
                if (h.bytes < 0)
                    throw std::runtime_error("gather: rank " + std::to_string(h.rank) + " reported a failure");
                seen[(std::size_t)h.rank] = true;
                data.resize((std::size_t)h.bytes);
                recvAll(open.back()->fd, data.data(), data.size());
                sink((int)h.rank, data.data(), data.size());
            }
            if (commit)
                commit();
        } catch (const std::exception& e) {
            failure = e.what();
        }
        const char status = failure.empty() ? 1 : 0;
        for (auto& c : open)
            (void)::send(c->fd, &status, 1, MSG_NOSIGNAL); // best effort: a sender that has gone does not matter any more
        if (!failure.empty())
            throw std::runtime_error(failure);
        return;
    }
    Socket c;
    std::string why;
    const GatherHeader h = { env.rank, (std::int64_t)bytes };
    if (!connectAndSend(env, h, buffer, timeoutSeconds, c, why))
        throw std::runtime_error(why);
    char status = 0;
    const ssize_t got = ::recv(c.fd, &status, 1, 0);
    if (got != 1)
        throw std::runtime_error("gather: rank 0 did not acknowledge the restart file within " + std::to_string(timeoutSeconds) + " s (it failed or left)");
    if (status != 1)
        throw std::runtime_error("gather: rank 0 reports that the restart file was NOT written");
}

void reportFailureToRankZero(const RankEnvironment& env, int timeoutSeconds) noexcept
{
    if (env.world <= 1 || env.rank == 0)
        return;
    try {
        Socket c;
        std::string why;
        const GatherHeader h = { env.rank, -1 };
        (void)connectAndSend(env, h, nullptr, gatherSeconds(timeoutSeconds), c, why);
    } catch (...) {
    }
}

} // namespace Nextsim
