// Fiber scheduler of the host emulator (see emu_runtime.h).  TEST INFRASTRUCTURE ONLY.
#include "emu_runtime.h"

#include <mutex>
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
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

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
std::vector<Fiber> fibers;
std::vector<WaveState> waves;
std::vector<char> stacks;
std::vector<char> dynsmem;
void* sched_sp = nullptr;
Fiber* cur = nullptr;
unsigned block_gen = 0;
int block_arrived = 0, block_alive = 0;
const std::function<void()>* body_fn = nullptr;
std::mutex launch_mu;

void yield() { emu_switch(&cur->sp, sched_sp); }

void trampoline() {
    (*body_fn)();
    cur->done = true;
    // a finished thread no longer takes part in barriers
    block_alive--;
    WaveState& w = waves[cur->wave];
    w.alive--;
    if (block_alive > 0 && block_arrived == block_alive) {
        block_arrived = 0;
        block_gen++;
    }
    if (w.alive > 0 && w.arrived == w.alive) {
        w.arrived = 0;
        w.gen++;
    }
    emu_switch(&cur->sp, sched_sp);
    std::abort();  // a finished fiber is never resumed
}
}  // namespace

void syncthreads() {
    block_arrived++;
    if (block_arrived == block_alive) {
        block_arrived = 0;
        block_gen++;
        return;
    }
    cur->wait = BLOCK;
    cur->wait_gen = block_gen;
    yield();
}

void wave_sync() {
    WaveState& w = waves[cur->wave];
    w.arrived++;
    if (w.arrived == w.alive) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    cur->wait = WAVE;
    cur->wait_gen = w.gen;
    yield();
}

uint64_t* wave_buf() { return waves[cur->wave].buf; }
int lane_id() { return cur->lane; }
int& lane_parity() { return cur->parity; }
int wave_lanes() { return waves[cur->wave].alive; }
char* dyn_smem() { return dynsmem.data(); }

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body) {
    std::lock_guard<std::mutex> lk(launch_mu);
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    if ((int)fibers.size() < nthreads) fibers.resize(nthreads);
    if (stacks.size() < (size_t)nthreads * kStack) stacks.resize((size_t)nthreads * kStack);
    if (dynsmem.size() < dyn_bytes + 64) dynsmem.resize(dyn_bytes + 64);
    waves.assign(nwaves, WaveState());
    g_blockDim = block;
    g_gridDim = grid;
    body_fn = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                block_gen = 0;
                block_arrived = 0;
                block_alive = nthreads;
                for (int w = 0; w < nwaves; ++w) {
                    waves[w].gen = 0;
                    waves[w].arrived = 0;
                    waves[w].alive = std::min(64, nthreads - 64 * w);
                }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.wait = NONE;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.wave = t / 64;
                    f.lane = t % 64;
                    f.parity = 0;
                    // initial frame: six callee-saved slots + return address = trampoline; the stack pointer at
                    // trampoline entry must be 8 (mod 16) as after a `call`
                    uintptr_t top = ((uintptr_t)(stacks.data() + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
                    void** frame = (void**)(top - 64);
                    for (int q = 0; q < 6; ++q) frame[q] = nullptr;
                    frame[6] = (void*)trampoline;
                    f.sp = (void*)frame;
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    bool progressed = false;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        if (f.wait == BLOCK && f.wait_gen == block_gen) continue;
                        if (f.wait == WAVE && f.wait_gen == waves[f.wave].gen) continue;
                        f.wait = NONE;
                        cur = &f;
                        g_threadIdx = f.tid;
                        emu_switch(&sched_sp, f.sp);
                        progressed = true;
                        if (f.done) remaining--;
                    }
                    if (!progressed) {
                        std::fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier/shuffle\n", bx, by, bz);
                        std::abort();
                    }
                }
            }
    body_fn = nullptr;
}
}  // namespace emu
