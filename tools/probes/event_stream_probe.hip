// event_stream_probe -- does the runtime follow an event back to the stream that recorded it?
//
//   record E on stream S, synchronise, destroy S; fill the host heap with zeroed blocks of the
//   sizes a freed stream object may have had; then hipStreamWaitEvent(T, E) / hipEventRecord(E, T) /
//   hipEventQuery(E) / hipEventDestroy(E) and look for blocks that are no longer zero.
//
// Why: round 4's long fuzz run caught a caller's array changing (one int32, 1 -> 2) inside
// trlda_batch_create, in the lookup of a recycled device allocation (hipEventQuery on the allocation's
// guard event; a watchpoint build saw the value change between the line before that loop and the line
// after it), when the guard had been recorded on the own stream of a model destroyed since.  With such
// guards destroyed together with the model's stream (trlda_hip.hip: purge_stream_guards, batch_settle,
// stream_alive) the write is gone.
// Result of THIS probe (ROCm 7.2, gfx950, profiles/r04_event_stream_probe.txt): no stray write in any
// of its twelve orders -- the stand-alone sequence does not reproduce what the library's did; the
// library's rule stays (an event is not touched once the stream that recorded it is gone).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<std::pair<unsigned char *, size_t>> g_blocks;

__global__ void touch(int *p) { if (p) p[threadIdx.x] = 1; }

static void fill_heap()
{
    for (size_t sz = 16; sz <= 16384; sz += 16)
        for (int rep = 0; rep < (sz <= 2048 ? 48 : 6); ++rep) {
            unsigned char *p = static_cast<unsigned char *>(std::malloc(sz));
            std::memset(p, 0, sz);
            g_blocks.push_back({p, sz});
        }
}

static int scan(const char *what)
{
    int hits = 0;
    for (auto &b : g_blocks)
        for (size_t i = 0; i < b.second; ++i)
            if (b.first[i]) {
                std::printf("  after %s: block of %zu bytes, offset %zu holds %d\n", what, b.second, i, b.first[i]);
                b.first[i] = 0;
                ++hits;
            }
    std::printf("%-44s %d byte(s) of freed-then-reused host memory written\n", what, hits);
    return hits;
}

int main()
{
    hipStream_t T;
    (void)hipStreamCreateWithFlags(&T, hipStreamNonBlocking);
    int total = 0;
    for (int variant = 0; variant < 3; ++variant)
    for (int trial = 0; trial < 4; ++trial) {
        hipStream_t S;
        hipEvent_t E;
        (void)hipStreamCreateWithFlags(&S, hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&E, hipEventDisableTiming);
        void *dbuf = nullptr, *hbuf = nullptr;
        (void)hipMalloc(&dbuf, 1 << 16);
        (void)hipHostMalloc(&hbuf, 1 << 16, hipHostMallocDefault);
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, S, static_cast<int *>(dbuf));
        (void)hipMemcpyAsync(hbuf, dbuf, 256, hipMemcpyDeviceToHost, S);
        // variant 0: record, synchronise, destroy; 1: record, destroy (which drains the stream);
        // 2: synchronise, record (nothing pending), synchronise, destroy -- the order of a batch closed
        // after its model's last call and before the model
        if (variant == 2)
            (void)hipStreamSynchronize(S);
        (void)hipEventRecord(E, S);
        if (variant != 1)
            (void)hipStreamSynchronize(S);
        (void)hipStreamDestroy(S);
        fill_heap();
        std::printf("variant %d trial %d\n", variant, trial);
        if (trial == 0) { (void)hipStreamWaitEvent(T, E, 0); (void)hipMemcpyAsync(dbuf, hbuf, 256, hipMemcpyHostToDevice, T); total += scan("hipStreamWaitEvent(other stream, E)"); }
        if (trial == 1) { (void)hipEventQuery(E); total += scan("hipEventQuery(E)"); }
        if (trial == 2) { (void)hipEventRecord(E, T); total += scan("hipEventRecord(E, other stream)"); }
        if (trial == 3) { (void)hipEventDestroy(E); total += scan("hipEventDestroy(E)"); }
        (void)hipStreamSynchronize(T);
        total += scan("  ... and a synchronisation");
        for (auto &b : g_blocks)
            std::free(b.first);
        g_blocks.clear();
    }
    std::printf("total %d\n", total);
    return 0;
}
