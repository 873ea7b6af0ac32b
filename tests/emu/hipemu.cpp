// Runtime of the single-threaded HIP emulator (TEST INFRASTRUCTURE ONLY),
// see tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

namespace hipemu {

static BlockState g_bs;
static const size_t kStack = 256 * 1024;

BlockState &bs() { return g_bs; }
void *dyn_smem() { return g_bs.dyn.data(); }

uint3_ tidx()
{
    int t = g_bs.cur;
    uint3_ r;
    r.x = t % g_bs.block.x;
    r.y = (t / g_bs.block.x) % g_bs.block.y;
    r.z = t / (g_bs.block.x * g_bs.block.y);
    return r;
}
uint3_ bidx_() { return g_bs.bidx; }
dim3 bdim() { return g_bs.block; }
dim3 gdim() { return g_bs.grid; }

// Context switch between the scheduler and the thread fibers.  glibc's
// swapcontext saves / restores the signal mask with a system call on every
// switch, which dominated the run time of barrier-heavy kernels (the LDS tile
// smoother: 1024 fibers x a dozen barriers per block); on x86-64 a switch of
// the callee-saved registers and the stack pointer is all that is needed.
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void **save_sp, void *new_sp);
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
static inline void to_sched(Fiber &f) { hipemu_switch(&f.sp, g_bs.sched_sp); }
static inline void to_fiber(Fiber &f) { hipemu_switch(&g_bs.sched_sp, f.sp); }
#else
static inline void to_sched(Fiber &f) { swapcontext(&f.ctx, &g_bs.sched); }
static inline void to_fiber(Fiber &f) { swapcontext(&g_bs.sched, &f.ctx); }
#endif

static void fiber_entry()
{
    (*g_bs.body)();
    Fiber &f = g_bs.fib[g_bs.cur];
    f.done = true;
    g_bs.nlive--;
    g_bs.wave_live[g_bs.cur >> 6]--;
    to_sched(f);   // never resumed
    __builtin_trap();
}

static void fiber_init(Fiber &f)
{
#ifdef HIPEMU_FAST_SWITCH
    // initial frame: six callee-saved registers, the entry point as the
    // return address of hipemu_switch, and a null caller above it, so that
    // rsp = 8 (mod 16) on entry like after a call
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;
    *--sp = (void *)fiber_entry;
    for (int k = 0; k < 6; k++) *--sp = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
}

void yield_()
{
    to_sched(g_bs.fib[g_bs.cur]);
}

void block_barrier()
{
    unsigned long gen = g_bs.bar_gen;
    g_bs.bar_arrived++;
    while (g_bs.bar_gen == gen) {
        if (g_bs.bar_arrived >= g_bs.nlive) {  // last one in releases everybody
            g_bs.bar_arrived = 0;
            g_bs.bar_gen++;
            break;
        }
        yield_();
    }
}

void wave_barrier()
{
    int w = g_bs.cur >> 6;
    unsigned long gen = g_bs.wave_gen[w];
    g_bs.wave_arrived[w]++;
    while (g_bs.wave_gen[w] == gen) {
        if (g_bs.wave_arrived[w] >= g_bs.wave_live[w]) {
            g_bs.wave_arrived[w] = 0;
            g_bs.wave_gen[w]++;
            break;
        }
        yield_();
    }
}

double shfl_read(double v, int src_lane)
{
    int w = g_bs.cur >> 6, lane = g_bs.cur & 63;
    g_bs.shfl[w * 64 + lane] = v;
    wave_barrier();
    double r = v;
    int src = w * 64 + src_lane;
    if (src_lane >= 0 && src_lane < 64 && src < g_bs.nthreads) r = g_bs.shfl[src];
    wave_barrier();
    return r;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
    BlockState &b = g_bs;
    b.grid = grid;
    b.block = block;
    b.nthreads = block.x * block.y * block.z;
    b.body = &body;
    b.dyn.assign(shmem + 64, 0);
    if ((int)b.fib.size() < b.nthreads) {
        size_t old = b.fib.size();
        b.fib.resize(b.nthreads);
        for (size_t k = old; k < b.fib.size(); k++) b.fib[k].stack = (char *)malloc(kStack);
    }
    int nw = (b.nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                b.bidx = {bx, by, bz};
                b.nlive = b.nthreads;
                b.bar_arrived = 0;
                b.wave_arrived.assign(nw, 0);
                b.wave_gen.assign(nw, 0);
                b.wave_live.assign(nw, 0);
                b.shfl.assign((size_t)nw * 64, 0.0);
                for (int t = 0; t < b.nthreads; t++) {
                    Fiber &f = b.fib[t];
                    f.done = false;
                    b.wave_live[t >> 6]++;
                    fiber_init(f);
                }
                while (b.nlive > 0)
                    for (int t = 0; t < b.nthreads; t++) {
                        if (b.fib[t].done) continue;
                        b.cur = t;
                        to_fiber(b.fib[t]);
                    }
            }
}

}  // namespace hipemu
