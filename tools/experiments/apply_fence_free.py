"""one-off source edit (kept for the record): agent-scope fences of the split-stream hand-offs -> write-through stores + agent-scope loads"""
import sys
p = sys.argv[1]
s = open(p).read()


def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)


rep('''    if (part < nparts - 1) {                                     // nobody reads the last range's function
        ps.bf[part * kWave + lane] = fn;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&ps.tick[part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }''', '''    // Hand-off without fences: the 64 words go out as agent-scope (write-through) stores, are drained, then the flag; the
    // readers use agent-scope loads.  An agent-scope release / acquire pair writes back and invalidates the XCD's whole L2
    // -- measured +4.6 us on the decode launch of 64 256x256 images with ONE stream split in two.
    if (part < nparts - 1) {                                     // nobody reads the last range's function
        __hip_atomic_store(&ps.bf[part * kWave + lane], fn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ps.tick[part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }''')
rep('''                while (__hip_atomic_load(&ps.tick[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint32_t r[kDecPartsMax - 1];
#pragma unroll
        for (int g = 0; g < kDecPartsMax - 1; ++g) r[g] = g < part ? ps.bf[g * kWave + lane] : 0u;''', '''                while (__hip_atomic_load(&ps.tick[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        uint32_t r[kDecPartsMax - 1];
#pragma unroll
        for (int g = 0; g < kDecPartsMax - 1; ++g)
            r[g] = g < part ? __hip_atomic_load(&ps.bf[g * kWave + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;''')
rep('''        unsigned int *d = x.tick + 4 * x.part;
        d[0] = count; d[1] = bits; d[2] = head;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&d[3], err ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);''', '''        // (agent-scope stores, drained, then the flag; agent-scope loads on the other side: no L2-wide fences, see part_exchange)
        unsigned int *d = x.tick + 4 * x.part;
        __hip_atomic_store(&d[0], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d[1], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d[2], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&d[3], err ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);''')
rep('''        while ((f = __hip_atomic_load(&d[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_desc[4 * tid] = d[0]; s_desc[4 * tid + 1] = d[1]; s_desc[4 * tid + 2] = d[2]; s_desc[4 * tid + 3] = f;''', '''        while ((f = __hip_atomic_load(&d[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        s_desc[4 * tid] = __hip_atomic_load(&d[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 1] = __hip_atomic_load(&d[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 2] = __hip_atomic_load(&d[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 3] = f;''')
open(p, 'w').write(s)
print("ok")
