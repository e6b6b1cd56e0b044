// TEST INFRASTRUCTURE ONLY — see include/hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <random>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hipemu {
static const size_t STACK = 256 * 1024;
// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer). glibc's swapcontext() makes a
// sigprocmask system call per switch, which dominates kernels that use many cross-lane operations.
extern "C" void knz_emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl knz_emu_switch
.type knz_emu_switch,@function
knz_emu_switch:
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
.size knz_emu_switch,.-knz_emu_switch
)");
struct Fiber { void* sp = nullptr; char* stack = nullptr; bool done = false; unsigned tid = 0; };
static std::vector<Fiber> fibers;
static void* schedSp = nullptr;
static int cur = -1;
static const std::function<void()>* bodyPtr = nullptr;
static unsigned blkArrived, blkGen, blkAlive;
static unsigned wvArrived[64], wvGen[64], wvAlive[64];
static uint64_t wvSlots[64][64];
static unsigned long progress = 0;

static void trampoline() {
    (*bodyPtr)();
    Fiber& f = fibers[cur];
    f.done = true;
    progress++;
    blkAlive--;
    wvAlive[f.tid >> 6]--;
    // a finished thread no longer takes part in barriers: release waiters if it was the last one missing
    if (blkAlive && blkArrived == blkAlive) { blkArrived = 0; blkGen++; }
    unsigned w = f.tid >> 6;
    if (wvAlive[w] && wvArrived[w] == wvAlive[w]) { wvArrived[w] = 0; wvGen[w]++; }
    knz_emu_switch(&f.sp, schedSp);
    abort();   // a finished fiber is never resumed
}
static inline void yield() { knz_emu_switch(&fibers[cur].sp, schedSp); }
void spin_pause() { progress++; yield(); }       // a polling loop lets the other fibers run (kernels bound their own spins)

void block_barrier() {
    unsigned g = blkGen;
    progress++;
    if (++blkArrived == blkAlive) { blkArrived = 0; blkGen++; return; }
    while (blkGen == g) yield();
}
void wave_barrier() {
    unsigned w = fibers[cur].tid >> 6;
    unsigned g = wvGen[w];
    progress++;
    if (++wvArrived[w] == wvAlive[w]) { wvArrived[w] = 0; wvGen[w]++; return; }
    while (wvGen[w] == g) yield();
}
uint64_t* wave_slots() { return wvSlots[fibers[cur].tid >> 6]; }
int lane() { return (int)(fibers[cur].tid & 63); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || nt > 1024) { fprintf(stderr, "hipemu: bad block size %u\n", nt); abort(); }
    static unsigned long launchNo = 0;
    launchNo++;
    if (getenv("KNZ_EMU_TRACE")) fprintf(stderr, "hipemu: launch #%lu grid %u x %u block %u\n", launchNo, grid.x, grid.y, nt);
    const char* mode = getenv("KNZ_EMU_SCHED");
    int sched = 0; // 0 fwd, 1 rev, 2 random
    if (mode && !strcmp(mode, "rev")) sched = 1;
    if (mode && !strcmp(mode, "rand")) sched = 2;
    static std::mt19937 rng(12345);
    if (fibers.size() < nt) fibers.resize(nt);
    for (unsigned i = 0; i < nt; i++) if (!fibers[i].stack) fibers[i].stack = (char*)malloc(STACK);
    bodyPtr = &body;
    blockDim = block; gridDim = grid;
    std::vector<unsigned> order(nt);
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        blkArrived = 0; blkGen = 0; blkAlive = nt;
        for (unsigned w = 0; w < 64; w++) { wvArrived[w] = 0; wvGen[w] = 0; wvAlive[w] = 0; }
        for (unsigned i = 0; i < nt; i++) {
            Fiber& f = fibers[i];
            f.done = false; f.tid = i;
            wvAlive[i >> 6]++;
            // initial frame: 6 callee-saved registers, then the entry point as the address `ret` jumps to
            uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
            void** spp = (void**)(top - 64);
            for (int r = 0; r < 6; r++) spp[r] = nullptr;
            spp[6] = (void*)trampoline;
            spp[7] = nullptr;
            f.sp = (void*)spp;
            order[i] = i;
        }
        if (sched == 1) std::reverse(order.begin(), order.end());
        unsigned remaining = nt;
        unsigned long lastProgress = progress; int stalls = 0;
        while (remaining) {
            if (sched == 2) std::shuffle(order.begin(), order.end(), rng);
            for (unsigned k = 0; k < nt; k++) {
                unsigned i = order[k];
                Fiber& f = fibers[i];
                if (f.done) continue;
                cur = (int)i;
                unsigned lin = i;
                threadIdx.x = lin % block.x; threadIdx.y = (lin / block.x) % block.y; threadIdx.z = lin / (block.x * block.y);
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                knz_emu_switch(&schedSp, f.sp);
                if (f.done) remaining--;
            }
            if (progress == lastProgress) {
                if (++stalls > 2) {
                    fprintf(stderr, "hipemu: deadlock (divergent barrier / cross-lane op) in block %u, %u of %u threads left, block barrier %u/%u, stuck:", bx, remaining, nt, blkArrived, blkAlive);
                    for (unsigned i = 0, shown = 0; i < nt && shown < 12; i++) if (!fibers[i].done) { fprintf(stderr, " %u", i); shown++; }
                    for (unsigned w = 0; w < (nt + 63) / 64; w++) fprintf(stderr, " [wave %u: %u/%u]", w, wvArrived[w], wvAlive[w]);
                    fprintf(stderr, "\n");
                    abort();
                }
            }
            else { stalls = 0; lastProgress = progress; }
        }
    }
    cur = -1;
}
} // namespace hipemu
