// Fiber scheduler behind tests/simt_emu/hip/hip_runtime.h (test infrastructure only).
#include <sys/mman.h>

#include "hip/hip_runtime.h"

namespace simt_emu {

Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {

enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int state = DONE;
    Idx tid{};
};

constexpr size_t STACK_SIZE = 512 * 1024;

std::vector<Fiber> g_fibers;
void *g_sched_sp = nullptr;
int g_cur = -1;
const std::function<void()> *g_body = nullptr;
unsigned long long g_scratch[16][64];   // per wave deposit
unsigned long long g_snapshot[16][64];  // per wave released values
unsigned long long g_active[16];

extern "C" void simt_emu_switch(void **from_sp, void *to_sp);
asm(R"(
.text
.globl simt_emu_switch
.type simt_emu_switch,@function
simt_emu_switch:
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
.size simt_emu_switch,.-simt_emu_switch
)");

void yield_to_scheduler(int new_state) {
    Fiber &f = g_fibers[g_cur];
    f.state = new_state;
    simt_emu_switch(&f.sp, g_sched_sp);
}

extern "C" void simt_emu_trampoline() {
    (*g_body)();
    yield_to_scheduler(DONE);
    std::fprintf(stderr, "simt_emu: resumed a finished fiber\n");
    std::abort();
}

void prepare_fiber(Fiber &f) {
    if (!f.stack) {
        f.stack = (char *)mmap(nullptr, STACK_SIZE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char *)MAP_FAILED) {
            std::perror("simt_emu mmap");
            std::abort();
        }
    }
    uintptr_t top = ((uintptr_t)f.stack + STACK_SIZE) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                             // fake return address of the trampoline's caller
    *--sp = (void *)&simt_emu_trampoline;        // `ret` target of the first switch
    for (int i = 0; i < 6; i++) *--sp = nullptr; // rbp rbx r12 r13 r14 r15
    f.sp = (void *)sp;
    f.state = RUNNABLE;
}

void resume(int i) {
    g_cur = i;
    g_threadIdx = g_fibers[i].tid;
    simt_emu_switch(&g_sched_sp, g_fibers[i].sp);
    g_cur = -1;
}

void run_block(unsigned n_threads) {
    unsigned n_waves = (n_threads + 63u) / 64u;
    if (n_waves > 16) {
        std::fprintf(stderr, "simt_emu: workgroup too large\n");
        std::abort();
    }
    for (;;) {
        for (unsigned i = 0; i < n_threads; i++)
            if (g_fibers[i].state == RUNNABLE) resume((int)i);
        // every fiber is now blocked or done
        bool released = false, any_block_wait = false, any_live = false;
        for (unsigned w = 0; w < n_waves; w++) {
            unsigned lo = w * 64u, hi = std::min(lo + 64u, n_threads);
            unsigned n_wave_wait = 0, n_block_wait = 0;
            unsigned long long mask = 0;
            for (unsigned i = lo; i < hi; i++) {
                if (g_fibers[i].state == WAIT_WAVE) {
                    n_wave_wait++;
                    mask |= 1ull << (i - lo);
                } else if (g_fibers[i].state == WAIT_BLOCK) {
                    n_block_wait++;
                }
            }
            if (n_wave_wait + n_block_wait > 0) any_live = true;
            if (n_block_wait) any_block_wait = true;
            if (n_wave_wait && n_block_wait) {
                std::fprintf(stderr,
                             "simt_emu: wave %u of block (%u,%u) has lanes at a wave collective and lanes at __syncthreads "
                             "(divergent rendezvous)\n",
                             w, g_blockIdx.x, g_blockIdx.y);
                std::abort();
            }
            if (n_wave_wait) {
                std::memcpy(g_snapshot[w], g_scratch[w], sizeof g_scratch[w]);
                g_active[w] = mask;
                for (unsigned i = lo; i < hi; i++)
                    if (g_fibers[i].state == WAIT_WAVE) g_fibers[i].state = RUNNABLE;
                released = true;
            }
        }
        if (released) continue;
        if (any_block_wait) {
            for (unsigned i = 0; i < n_threads; i++)
                if (g_fibers[i].state == WAIT_BLOCK) g_fibers[i].state = RUNNABLE;
            continue;
        }
        if (!any_live) return;
    }
}

}  // namespace

void barrier() { yield_to_scheduler(WAIT_BLOCK); }

int wave_exchange(unsigned long long v, unsigned long long *vals, unsigned long long *active) {
    unsigned lin = g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z);
    unsigned w = lin >> 6, lane = lin & 63u;
    g_scratch[w][lane] = v;
    yield_to_scheduler(WAIT_WAVE);
    std::memcpy(vals, g_snapshot[w], sizeof(unsigned long long) * 64);
    *active = g_active[w];
    return (int)lane;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    unsigned n_threads = block.x * block.y * block.z;
    if (g_fibers.size() < n_threads) g_fibers.resize(n_threads);
    g_body = &body;
    g_blockDim = Idx{block.x, block.y, block.z};
    g_gridDim = Idx{grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g_blockIdx = Idx{bx, by, bz};
                for (unsigned t = 0; t < n_threads; t++) {
                    Fiber &f = g_fibers[t];
                    f.tid = Idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    prepare_fiber(f);
                }
                run_block(n_threads);
            }
    g_body = nullptr;
}

}  // namespace simt_emu
