// Fiber scheduler of the host emulator (see emu_runtime.h).  TEST INFRASTRUCTURE ONLY.
#include "emu_runtime.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer).  glibc's swapcontext() issues a
// sigprocmask system call per switch, which made shuffle-heavy kernels ~50x slower to emulate.
#if !defined(__x86_64__)
#error "the host emulator's context switch is written for x86-64"
#endif
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace emu {
// Workgroups are independent, so a launch spreads them over OS threads (OpenMP, dynamic schedule); every fiber of one workgroup
// runs on the thread that picked the workgroup up, all scheduler state is thread-local.  Global-memory atomics of the kernels are
// real atomics (emu_runtime.h), `__shared__` arrays are thread-local statics.  CAD_EMU_THREADS=1 restores the serial schedule.
thread_local dim3 g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;
enum Wait { NONE = 0, BLOCK = 1, WAVE = 2 };
struct Fiber {
    void* sp = nullptr;
    dim3 tid;
    bool done = false;
    int wait = NONE;
    unsigned wait_gen = 0;
    int wave = 0, lane = 0, parity = 0;
};
struct WaveState {
    unsigned gen = 0;
    int arrived = 0, alive = 0;
    uint64_t buf[128];
};
// per OS thread: the workgroup it is running
struct Worker {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    char* stacks = nullptr;  // mmap'ed, touched lazily (a fiber uses a few KB of its 256 KB)
    size_t stacks_bytes = 0;
    std::vector<char> dynsmem;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    unsigned block_gen = 0;
    int block_arrived = 0, block_alive = 0;
    ~Worker() {
        if (stacks) munmap(stacks, stacks_bytes);
    }
};
thread_local Worker tw;
const std::function<void()>* body_fn = nullptr;
std::mutex launch_mu;

void yield() { emu_switch(&tw.cur->sp, tw.sched_sp); }

void trampoline() {
    (*body_fn)();
    Worker& w_ = tw;
    w_.cur->done = true;
    // a finished thread no longer takes part in barriers
    w_.block_alive--;
    WaveState& w = w_.waves[w_.cur->wave];
    w.alive--;
    if (w_.block_alive > 0 && w_.block_arrived == w_.block_alive) {
        w_.block_arrived = 0;
        w_.block_gen++;
    }
    if (w.alive > 0 && w.arrived == w.alive) {
        w.arrived = 0;
        w.gen++;
    }
    emu_switch(&w_.cur->sp, w_.sched_sp);
    std::abort();  // a finished fiber is never resumed
}

int max_threads() {
    static const int n = [] {
        const char* e = std::getenv("CAD_EMU_THREADS");
        int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(v, 32));
    }();
    return n;
}

void run_block(unsigned bx, unsigned by, unsigned bz, dim3 block, size_t dyn_bytes) {
    Worker& W = tw;
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    if ((int)W.fibers.size() < nthreads) W.fibers.resize(nthreads);
    if (W.stacks_bytes < (size_t)nthreads * kStack) {
        if (W.stacks) munmap(W.stacks, W.stacks_bytes);
        W.stacks_bytes = (size_t)nthreads * kStack;
        void* m = mmap(nullptr, W.stacks_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) {
            std::fprintf(stderr, "emu: cannot map %zu bytes of fiber stacks\n", W.stacks_bytes);
            std::abort();
        }
        W.stacks = (char*)m;
    }
    if (W.dynsmem.size() < dyn_bytes + 64) W.dynsmem.resize(dyn_bytes + 64);
    if ((int)W.waves.size() < nwaves) W.waves.resize(nwaves);
    g_blockIdx = dim3(bx, by, bz);
    W.block_gen = 0;
    W.block_arrived = 0;
    W.block_alive = nthreads;
    for (int w = 0; w < nwaves; ++w) {
        W.waves[w].gen = 0;
        W.waves[w].arrived = 0;
        W.waves[w].alive = std::min(64, nthreads - 64 * w);
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = W.fibers[t];
        f.done = false;
        f.wait = NONE;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.wave = t / 64;
        f.lane = t % 64;
        f.parity = 0;
        // initial frame: six callee-saved slots + return address = trampoline; the stack pointer at
        // trampoline entry must be 8 (mod 16) as after a `call`
        uintptr_t top = ((uintptr_t)(W.stacks + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
        void** frame = (void**)(top - 64);
        for (int q = 0; q < 6; ++q) frame[q] = nullptr;
        frame[6] = (void*)trampoline;
        f.sp = (void*)frame;
    }
    int remaining = nthreads;
    while (remaining > 0) {
        bool progressed = false;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = W.fibers[t];
            if (f.done) continue;
            if (f.wait == BLOCK && f.wait_gen == W.block_gen) continue;
            if (f.wait == WAVE && f.wait_gen == W.waves[f.wave].gen) continue;
            f.wait = NONE;
            W.cur = &f;
            g_threadIdx = f.tid;
            emu_switch(&W.sched_sp, f.sp);
            progressed = true;
            if (f.done) remaining--;
        }
        if (!progressed) {
            std::fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier/shuffle\n", bx, by, bz);
            std::abort();
        }
    }
}
}  // namespace

void syncthreads() {
    Worker& W = tw;
    W.block_arrived++;
    if (W.block_arrived == W.block_alive) {
        W.block_arrived = 0;
        W.block_gen++;
        return;
    }
    W.cur->wait = BLOCK;
    W.cur->wait_gen = W.block_gen;
    yield();
}

void wave_sync() {
    Worker& W = tw;
    WaveState& w = W.waves[W.cur->wave];
    w.arrived++;
    if (w.arrived == w.alive) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    W.cur->wait = WAVE;
    W.cur->wait_gen = w.gen;
    yield();
}

uint64_t* wave_buf() { return tw.waves[tw.cur->wave].buf; }
int lane_id() { return tw.cur->lane; }
int& lane_parity() { return tw.cur->parity; }
int wave_lanes() { return tw.waves[tw.cur->wave].alive; }
char* dyn_smem() { return tw.dynsmem.data(); }

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body) {
    std::lock_guard<std::mutex> lk(launch_mu);
    g_blockDim = block;
    g_gridDim = grid;
    body_fn = &body;
    const long total = (long)grid.x * grid.y * grid.z;
    const int nthr = (int)std::min<long>(max_threads(), total);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr) if (nthr > 1)
    for (long b = 0; b < total; ++b) {
        const unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y), bz = (unsigned)(b / ((long)grid.x * grid.y));
        run_block(bx, by, bz, block, dyn_bytes);
    }
    body_fn = nullptr;
}
}  // namespace emu
