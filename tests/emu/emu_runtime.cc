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

// ThreadSanitizer build (make tsan): every GPU thread is a TSan fiber, the scheduler switches WITHOUT a synchronisation
// edge, and the only happens-before edges are the ones the machine gives: launch boundaries, block barriers, and the
// rendezvous of the lanes at a wave-level operation.  Two plain accesses of the same location from different threads
// with none of these in between -- inside a block or across the blocks of a launch -- are reported as a data race.
// This file itself is compiled WITHOUT instrumentation (-DHIPEMU_TSAN only): the scheduler's bookkeeping is not kernel data.
#ifdef HIPEMU_TSAN
extern "C" {
// The runtime's interface for managed heaps is the one public way to make it forget the access history of a range that is
// not malloc'ed: the section of the __shared__ statics is registered as such a heap and 're-allocated' at every block start.
void __tsan_java_init(unsigned long heap_begin, unsigned long heap_size);
void __tsan_java_alloc(unsigned long ptr, unsigned long size);
void __tsan_java_free(unsigned long ptr, unsigned long size);
void *__tsan_get_current_fiber(void);
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);
void __tsan_acquire(void *addr);
void __tsan_release(void *addr);
}
#define TSAN_ONLY(x) x
#else
#define TSAN_ONLY(x)
#endif

namespace hipemu {

thread_local ThreadCtx *g_ctx = nullptr;

namespace {
enum State : uint8_t { RUNNABLE, AT_WAVE_OP, AT_BARRIER, SLEEPING, DONE };  // SLEEPING: inside s_sleep (a spin-wait): resumed after the other waves have run
struct Fiber {
    void *sp = nullptr;  // saved stack pointer while the fiber is not running
    State state = DONE;
    const void *site = nullptr;
    uint64_t val = 0;
    Snap *out = nullptr;
    ThreadCtx ctx;
    void *tsan = nullptr, *tsan_pool[3] = {nullptr, nullptr, nullptr};
    int stack_slot = 0;
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
    size_t lds_bytes = 0;
    unsigned long block_counter = 0;  // blocks run so far, over all launches
    void *sched_tsan = nullptr;
    // addresses of the happens-before edges.  The barrier and wave objects of a block come from a pool and are never shared
    // with another block of the launch: a shared object would order the blocks through their barriers
    char launch_sync = 0, done_sync = 0;
    static constexpr size_t SYNC_POOL = 1 << 20;
    std::vector<char> sync_pool;
    size_t sync_next = 0;
    char *bar_sync = nullptr, *wave_sync[MAX_THREADS / WAVE] = {nullptr};
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

#ifdef HIPEMU_TSAN
// Drop the race detector's access history of a range (memory that changes hands WITHOUT a happens-before edge: the LDS of the
// next block, the stack of the next thread).  The runtime's managed-heap interface is the one public way to do that for
// memory that is not malloc'ed; the "heap" registered with it is the whole address space.
constexpr size_t STACK_WINDOW = 8 << 10;   // the part of a fiber stack the kernels can reach (measured: < 2 KB); checked at every switch
void forget(const void *p, size_t n) {
    static bool registered = false;
    if (!registered) {
        __tsan_java_init(8ul, (1ul << 47) - 16);
        registered = true;
    }
    unsigned long lo = (unsigned long)p & ~7ul, hi = ((unsigned long)p + n + 7) & ~7ul;  // whole 8-byte cells (allocations are 16-byte granular)
    if (hi > lo) {
        __tsan_java_free(lo, hi - lo);
        __tsan_java_alloc(lo, hi - lo);
    }
}
#endif

void yield_to_scheduler() {
    Machine *m = g_m;
    Fiber *f = m->cur;
#ifdef HIPEMU_TSAN
    {
        const uintptr_t top = (uintptr_t)(m->stacks + (size_t)(f->stack_slot + 1) * STACK_BYTES), here = (uintptr_t)__builtin_frame_address(0);
        if (top - here > STACK_WINDOW - 1024) {
            std::fprintf(stderr, "hipemu: a kernel thread uses more stack than the race detector's window\n");
            std::abort();
        }
    }
#endif
    TSAN_ONLY(__tsan_switch_to_fiber(m->sched_tsan, 1);)
    hipemu_switch(&f->sp, m->sched_sp);
    g_ctx = &f->ctx;  // resumed
}

extern "C" void hipemu_fiber_main() {
    Machine *m = g_m;
    Fiber *f = m->cur;
    g_ctx = &f->ctx;
    TSAN_ONLY(__tsan_acquire(&m->launch_sync);)  // everything the host did before the launch
    m->tramp(m->closure);
    TSAN_ONLY(__tsan_release(&m->done_sync);)
    f->state = DONE;
    TSAN_ONLY(__tsan_switch_to_fiber(m->sched_tsan, 1);)
    hipemu_switch(&f->sp, m->sched_sp);
    std::abort();  // a finished fiber is never resumed
}

void prepare(Machine *m, int t, int stack_slot) {
    Fiber &f = m->fibers[t];
    f.stack_slot = stack_slot;
    uintptr_t top = (uintptr_t)(m->stacks + (size_t)(stack_slot + 1) * STACK_BYTES);
    top &= ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                         // keeps the entry frame 16-byte aligned (as after a call)
    *--sp = (void *)&hipemu_fiber_main;      // popped by `ret`
    for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.state = RUNNABLE;
#ifdef HIPEMU_TSAN
    // The detector sees the accesses of one of its fibers as program-ordered, whoever ran on it.  Thread t of consecutive
    // blocks must therefore never share a detector fiber; three generations are kept and used in turn (creating and
    // destroying one per GPU thread costs ~60 us each, which made the race-detecting build unusable).
    void *&slot = f.tsan_pool[m->block_counter % 3];
    if (!slot) slot = __tsan_create_fiber(0);
    f.tsan = slot;
    forget((const void *)(top - STACK_WINDOW), STACK_WINDOW);
#endif
}

// word-by-word through volatile: no memcpy call, which a sanitizer runtime would intercept and attribute to the scheduler
void deliver(Snap *to, const Snap &from) {
    volatile uint64_t *d = reinterpret_cast<volatile uint64_t *>(to);
    const uint64_t *s = reinterpret_cast<const uint64_t *>(&from);
    for (size_t i = 0; i < sizeof(Snap) / sizeof(uint64_t); ++i) d[i] = s[i];
}

void run(Machine *m, Fiber &f) {
    m->cur = &f;
    TSAN_ONLY(__tsan_switch_to_fiber(f.tsan, 1);)  // 1 = no synchronisation between the two fibers
    hipemu_switch(&m->sched_sp, f.sp);
    m->cur = nullptr;
}

extern "C" char __start_hipemu_lds[] __attribute__((weak)), __stop_hipemu_lds[] __attribute__((weak));  // the __shared__ statics

void run_block(Machine *m, int n_threads) {
#ifdef HIPEMU_TSAN
    // Every block has its own LDS on the real machine: what earlier blocks did to these bytes is not a conflict.  The
    // history of the static section and of the dynamic carve is dropped, then the block's threads are given everything the
    // scheduler (= the host) has done so far -- and nothing any other GPU thread of this launch has done.
    {
        if (&__start_hipemu_lds[0] && &__stop_hipemu_lds[0] > &__start_hipemu_lds[0]) forget(__start_hipemu_lds, (size_t)(&__stop_hipemu_lds[0] - &__start_hipemu_lds[0]));
        if (m->lds_bytes) forget(m->lds, m->lds_bytes);
        __tsan_release(&m->launch_sync);
        if (m->sync_pool.empty()) m->sync_pool.assign(Machine::SYNC_POOL, 0);
        auto next = [&]() { return &m->sync_pool[m->sync_next++ % Machine::SYNC_POOL]; };
        m->bar_sync = next();
        for (int w = 0; w < MAX_THREADS / WAVE; ++w) m->wave_sync[w] = next();
    }
#endif
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
                // a lane that spins in s_sleep holds its wavefront where it is (on the machine the other lanes are masked off
                // until it leaves the loop): the wave's pending operations wait, the other wavefronts get their turn
                bool sleeping = false;
                for (int t = l0; t < l1; ++t) sleeping |= m->fibers[t].state == SLEEPING;
                if (sleeping) break;
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
                        deliver(m->fibers[t].out, s);
                        m->fibers[t].state = RUNNABLE;
                    }
            }
        }
        // lanes that slept go on (what they wait for may have happened meanwhile); the barrier stays shut until nobody sleeps
        {
            int slept = 0;
            for (int t = 0; t < n_threads; ++t)
                if (m->fibers[t].state == SLEEPING) {
                    m->fibers[t].state = RUNNABLE;
                    ++slept;
                }
            if (slept) continue;
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
    Machine *m = g_m;
    Fiber *f = m->cur;
    f->state = AT_WAVE_OP;
    f->site = site;
    f->val = my_val;
    f->out = &out;
    TSAN_ONLY(char *ws = m->wave_sync[(f - m->fibers.data()) / WAVE]; __tsan_release(ws);)
    yield_to_scheduler();
    TSAN_ONLY(__tsan_acquire(ws);)  // the lanes of a wavefront are in lockstep at a wave-level operation
}

void sleep_yield() {  // s_sleep: the calling lane spins on something another wavefront of the block will do
    g_m->cur->state = SLEEPING;
    yield_to_scheduler();
}

void block_barrier() {
    Machine *m = g_m;
    m->cur->state = AT_BARRIER;
    TSAN_ONLY(__tsan_release(m->bar_sync);)
    yield_to_scheduler();
    TSAN_ONLY(__tsan_acquire(m->bar_sync);)
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
    void *dyn = nullptr;  // dynamic LDS, exactly the requested size: an address-sanitizer build sees overruns of the carve
    if (posix_memalign(&dyn, 16, lds ? lds : 1) != 0) std::abort();
    m->lds = static_cast<unsigned char *>(dyn);
    m->lds_bytes = lds;
    m->tramp = tramp;
    m->closure = closure;
    Machine *outer = g_m;
    ThreadCtx *outer_ctx = g_ctx;
    g_m = m;
    TSAN_ONLY(m->sched_tsan = __tsan_get_current_fiber();)
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                for (int t = 0; t < n_threads; ++t) {
                    prepare(m, t, t);
                    const unsigned ut = (unsigned)t;  // linear thread id: x fastest, wavefronts of 64 consecutive ids
                    m->fibers[t].ctx = ThreadCtx{dim3(ut % block.x, ut / block.x % block.y, ut / (block.x * block.y)), dim3(bx, by, bz), block, grid};
                }
                for (size_t i = 0; i < lds; ++i) static_cast<volatile unsigned char *>(dyn)[i] = 0;  // not memset: see deliver()
                run_block(m, n_threads);
                ++m->block_counter;
            }
    TSAN_ONLY(__tsan_acquire(&m->done_sync);)  // a launch is complete when the host continues
    g_m = outer;
    g_ctx = outer_ctx;
    m->lds = nullptr;
    std::free(dyn);
}

}  // namespace hipemu
