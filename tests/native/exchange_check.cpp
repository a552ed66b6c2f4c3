// tests/native/exchange_check.cpp -- lz::exchange_run under recording / failing stubs (no GPU, no RCCL): the group of
// lanczos_resample_multi_root must look at every code, stop queueing at the first failure and ALWAYS close an opened group.
#include <cstdio>
#include <string>
#include <vector>

#include "lanczos_exchange.hpp"

struct Stub {
    std::vector<std::string> log;
    int fail_start = 0, fail_end = 0, fail_send_at = -1, fail_recv_at = -1, sends = 0, recvs = 0, open = 0;
    int group_start() {
        log.push_back("start");
        if (fail_start) return fail_start;
        open++;
        return 0;
    }
    int group_end() {
        log.push_back("end");
        open--;
        return fail_end;
    }
    int send(int rank, size_t off, size_t bytes, int peer) {
        char b[128];
        snprintf(b, sizeof b, "send r%d+%zu %zu -> %d", rank, off, bytes, peer);
        log.push_back(b);
        return sends++ == fail_send_at ? 7 : 0;
    }
    int recv(int rank, size_t off, size_t bytes, int peer) {
        char b[128];
        snprintf(b, sizeof b, "recv r%d+%zu %zu <- %d", rank, off, bytes, peer);
        log.push_back(b);
        return recvs++ == fail_recv_at ? 9 : 0;
    }
};

static int fails = 0;
#define EXPECT(c)                                               \
    do {                                                        \
        if (!(c)) {                                             \
            printf("FAILED line %d: %s\n", __LINE__, #c);       \
            fails++;                                            \
        }                                                       \
    } while (0)

int main() {
    using namespace lz;
    ExGeometry g{1000, 4000, 10, 20, 3, true};
    std::vector<ExShare> sh(3);
    sh[0].r0 = 0, sh[0].rows = 80, sh[0].i0 = 0, sh[0].irows = 44, sh[0].in_bytes = 3 * 440, sh[0].out_bytes = 3 * 1600;
    sh[1].r0 = 80, sh[1].rows = 60, sh[1].i0 = 37, sh[1].irows = 36, sh[1].in_bytes = 3 * 360, sh[1].out_bytes = 3 * 1200;
    sh[2].r0 = 140, sh[2].rows = 60, sh[2].i0 = 67, sh[2].irows = 33, sh[2].in_bytes = 3 * 330, sh[2].out_bytes = 3 * 1200;
    std::vector<ExXfer> sc, ga;
    exchange_scatter_plan(g, sh, &sc);
    exchange_gather_plan(g, sh, &ga);
    EXPECT(sc.size() == 6 && ga.size() == 6);
    EXPECT(sc[0].src == 0 && sc[0].dst == 1 && sc[0].src_off == 370 && sc[0].dst_off == 0 && sc[0].bytes == 360);
    EXPECT(sc[4].dst == 2 && sc[4].src_off == 1000 + 670 && sc[4].dst_off == 330 && sc[4].bytes == 330);
    EXPECT(ga[5].src == 2 && ga[5].dst == 0 && ga[5].src_off == 2400 && ga[5].dst_off == 8000 + 2800 && ga[5].bytes == 1200);
    {   // clean run: start, (send, recv) per message in order, end
        Stub s;
        int at = 99;
        EXPECT(exchange_run(s, sc, &at) == 0 && at == -2);
        EXPECT(s.log.size() == 2 + 2 * sc.size() && s.log.front() == "start" && s.log.back() == "end" && s.open == 0);
        EXPECT(s.log[1] == "send r0+370 360 -> 1" && s.log[2] == "recv r1+0 360 <- 0");
    }
    {   // a send is refused: nothing more is queued, the group is closed, the send's code comes back
        Stub s;
        s.fail_send_at = 2;
        int at = 0;
        EXPECT(exchange_run(s, sc, &at) == 7 && at == 2);
        EXPECT(s.sends == 3 && s.recvs == 2 && s.log.back() == "end" && s.open == 0);
    }
    {   // a recv is refused
        Stub s;
        s.fail_recv_at = 0;
        int at = 0;
        EXPECT(exchange_run(s, ga, &at) == 9 && at == 0);
        EXPECT(s.sends == 1 && s.recvs == 1 && s.log.back() == "end" && s.open == 0);
    }
    {   // the group cannot be opened: nothing is queued, nothing is closed
        Stub s;
        s.fail_start = 3;
        int at = 0;
        EXPECT(exchange_run(s, sc, &at) == 3 && at == -1);
        EXPECT(s.log.size() == 1 && s.open == 0);
    }
    {   // the group's end reports the failure (RCCL queues lazily: errors often surface here)
        Stub s;
        s.fail_end = 5;
        int at = 0;
        EXPECT(exchange_run(s, sc, &at) == 5 && at == (int)sc.size());
        EXPECT(s.open == 0);
    }
    {   // a send fails AND the end fails: the first failure is the one reported
        Stub s;
        s.fail_send_at = 0;
        s.fail_end = 5;
        int at = 0;
        EXPECT(exchange_run(s, sc, &at) == 7 && at == 0 && s.open == 0);
    }
    {   // empty list (one device, or all peers' shares empty): an empty group is still balanced
        Stub s;
        std::vector<ExXfer> none;
        EXPECT(exchange_run(s, none, nullptr) == 0 && s.log.size() == 2 && s.open == 0);
    }
    printf(fails ? "exchange_run: %d FAILED\n" : "exchange_run: all cases ok\n", fails);
    return fails ? 1 : 0;
}
