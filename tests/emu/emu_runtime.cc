// TEST INFRASTRUCTURE -- the scheduler behind tests/emu/hip/hip_runtime.h: GPU threads as fibers.
//
// A kernel launch runs its blocks one after the other on the calling thread.  Inside a block every thread is a fiber
// with its own stack; a fiber runs until it reaches a wave-level operation, a block barrier or the end of the kernel.
// A wavefront is advanced until all of its lanes wait at the block barrier (or are finished):
//   * the lanes that wait at wave-level operations are grouped by call site (the return address inside the kernel, after
//     inlining); the group with the LOWEST address is completed first.  With code laid out in source order (-O0 / -O1)
//     this runs the body of a divergent region before the reconvergence point after it, as the hardware does;
//   * every member of the completed group receives the operands of all members (Snap) and continues.
// When every live thread of the block waits at the barrier, the barrier opens.  Nothing runs concurrently: races of the
// real machine are not reproduced, and nothing here measures time.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <cstdio>
#include <mutex>
#include <vector>

namespace hipemu {

thread_local ThreadCtx *g_ctx = nullptr;

namespace {
enum State : uint8_t { RUNNABLE, AT_WAVE_OP, AT_BARRIER, DONE };
struct Fiber {
    void *sp = nullptr;  // saved stack pointer while the fiber is not running
    State state = DONE;
    const void *site = nullptr;
    uint64_t val = 0;
    Snap *out = nullptr;
    ThreadCtx ctx;
};
constexpr size_t STACK_BYTES = 256 << 10;
constexpr int MAX_THREADS = 1024;

struct Machine {
    unsigned char *stacks = nullptr;  // MAX_THREADS stacks, mapped once
    std::vector<Fiber> fibers;
    void *sched_sp = nullptr;
    Fiber *cur = nullptr;
    void (*tramp)(void *) = nullptr;
    void *closure = nullptr;
    unsigned char *lds = nullptr;  // dynamic LDS of the running launch: an allocation of exactly the requested size
};
thread_local Machine *g_m = nullptr;
std::mutex g_launch_mutex;  // one emulated device: launches from different host threads take turns

extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

void yield_to_scheduler() {
    Machine *m = g_m;
    Fiber *f = m->cur;
    hipemu_switch(&f->sp, m->sched_sp);
    g_ctx = &f->ctx;  // resumed
}

extern "C" void hipemu_fiber_main() {
    Machine *m = g_m;
    Fiber *f = m->cur;
    g_ctx = &f->ctx;
    m->tramp(m->closure);
    f->state = DONE;
    hipemu_switch(&f->sp, m->sched_sp);
    std::abort();  // a finished fiber is never resumed
}

void prepare(Machine *m, int t) {
    Fiber &f = m->fibers[t];
    uintptr_t top = (uintptr_t)(m->stacks + (size_t)(t + 1) * STACK_BYTES);
    top &= ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                         // keeps the entry frame 16-byte aligned (as after a call)
    *--sp = (void *)&hipemu_fiber_main;      // popped by `ret`
    for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.state = RUNNABLE;
}

void run(Machine *m, Fiber &f) {
    m->cur = &f;
    hipemu_switch(&m->sched_sp, f.sp);
    m->cur = nullptr;
}

void run_block(Machine *m, int n_threads) {
    const int n_waves = (n_threads + WAVE - 1) / WAVE;
    int live = n_threads;
    while (live > 0) {
        for (int w = 0; w < n_waves; ++w) {
            const int l0 = w * WAVE, l1 = l0 + WAVE < n_threads ? l0 + WAVE : n_threads;
            while (true) {
                bool ran = false;
                for (int t = l0; t < l1; ++t)
                    if (m->fibers[t].state == RUNNABLE) {
                        run(m, m->fibers[t]);
                        if (m->fibers[t].state == DONE) --live;
                        ran = true;
                    }
                // complete the pending wave-level operation with the lowest call-site address
                const void *site = nullptr;
                for (int t = l0; t < l1; ++t)
                    if (m->fibers[t].state == AT_WAVE_OP && (!site || m->fibers[t].site < site)) site = m->fibers[t].site;
                if (!site) {
                    if (!ran) break;  // the whole wave waits at the barrier or is finished
                    continue;
                }
                Snap s;
                s.mask = 0;
                for (int l = 0; l < WAVE; ++l) s.val[l] = 0;
                for (int t = l0; t < l1; ++t)
                    if (m->fibers[t].state == AT_WAVE_OP && m->fibers[t].site == site) {
                        s.mask |= 1ull << (t - l0);
                        s.val[t - l0] = m->fibers[t].val;
                    }
                for (int t = l0; t < l1; ++t)
                    if (m->fibers[t].state == AT_WAVE_OP && m->fibers[t].site == site) {
                        *m->fibers[t].out = s;
                        m->fibers[t].state = RUNNABLE;
                    }
            }
        }
        // every live thread waits at the block barrier: open it
        int waiting = 0;
        for (int t = 0; t < n_threads; ++t)
            if (m->fibers[t].state == AT_BARRIER) {
                m->fibers[t].state = RUNNABLE;
                ++waiting;
            }
        if (live > 0 && waiting == 0) {
            std::fprintf(stderr, "hipemu: block cannot make progress (%d live threads)\n", live);
            std::abort();
        }
    }
}
}  // namespace

void wave_exchange(uint64_t my_val, const void *site, Snap &out) {
    Fiber *f = g_m->cur;
    f->state = AT_WAVE_OP;
    f->site = site;
    f->val = my_val;
    f->out = &out;
    yield_to_scheduler();
}

void block_barrier() {
    g_m->cur->state = AT_BARRIER;
    yield_to_scheduler();
}

unsigned char *dyn_shared() { return g_m->lds; }

void launch_impl(dim3 grid, dim3 block, size_t lds, void (*tramp)(void *), void *closure) {
    std::lock_guard<std::mutex> lk(g_launch_mutex);
    static thread_local Machine machine;
    Machine *m = &machine;
    if (!m->stacks) {
        void *p = mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) std::abort();
        m->stacks = (unsigned char *)p;
        m->fibers.resize(MAX_THREADS);
    }
    const int n_threads = (int)(block.x * block.y * block.z);
    if (n_threads > MAX_THREADS) {
        std::fprintf(stderr, "hipemu: unsupported block shape\n");
        std::abort();
    }
    void *dyn = nullptr;  // exactly the requested size: a sanitizer build sees overruns of the LDS carve
    if (posix_memalign(&dyn, 16, lds ? lds : 1) != 0) std::abort();
    std::memset(dyn, 0, lds);
    m->lds = static_cast<unsigned char *>(dyn);
    m->tramp = tramp;
    m->closure = closure;
    Machine *outer = g_m;
    ThreadCtx *outer_ctx = g_ctx;
    g_m = m;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                for (int t = 0; t < n_threads; ++t) {
                    prepare(m, t);
                    const unsigned ut = (unsigned)t;  // linear thread id: x fastest, wavefronts of 64 consecutive ids
                    m->fibers[t].ctx = ThreadCtx{dim3(ut % block.x, ut / block.x % block.y, ut / (block.x * block.y)), dim3(bx, by, bz), block, grid};
                }
                run_block(m, n_threads);
            }
    g_m = outer;
    g_ctx = outer_ctx;
    m->lds = nullptr;
    std::free(dyn);
}

}  // namespace hipemu
