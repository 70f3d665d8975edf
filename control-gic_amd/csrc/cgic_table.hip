// cgic_table.hip -- host-side static Huffman table (HuffmanCoding.__init__,
// make_heap, merge_nodes, make_codes: reference CGIC/tools/indices_coding.py:10-17,
// 46-75) plus the device images the coder kernels read.
//
// The reference's tree shape under frequency ties is an artefact of CPython's
// heapq (binary heap on a list, strict '<' on freq only): heappush appends and
// bubbles the new item up while it is < its parent; heappop moves the last
// item to the root, walks the hole down to a leaf always promoting the smaller
// child -- the RIGHT one unless left < right -- and then bubbles the item back
// up.  PyHeap below implements exactly those two operations; any other
// priority queue gives a different (equally optimal, but not bit-identical)
// code.  Also errors, the thread-local message, and ABI/version queries.
#include "cgic_common.h"

#include <stdarg.h>

#include <map>
#include <utility>
#include <mutex>
#include <new>
#include <vector>

namespace cgic {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return CGIC_ERR_HIP;
}

// ---- ticket pool ------------------------------------------------------------------------------
// Kernels that hand work to "the last workgroup to arrive" need counters that are zero when the
// launch starts.  A memset node per launch costs ~5 us as a fill kernel, so the library owns
// zero-initialised device memory and every user resets its counter when it is done:
//  * eager launches take the next slots of a ring that belongs to THEIR STREAM (launches of one stream run in order, so a
//    slot is reused only by a launch that starts after its previous user has ended and reset it; with one ring for all
//    streams, two streams with large batches in flight could be handed the same slots -- round-2 advisor finding);
//  * launches being captured into a hipGraph take slots of a per-device pool that stay theirs (the graph may be replayed
//    at any time later) until the caller says the graph is gone: captures made between cgic_ticket_scope_begin() and
//    cgic_ticket_scope_end() on a thread are tagged with the scope's id and cgic_ticket_scope_release(id) returns their
//    slots (pipeline.LaneStream / GraphLanes do this when a graph object is destroyed).  Captures outside a scope keep
//    their slots for the life of the process, as before.
// Pools and rings are created by EAGER calls (allocation is illegal in capture): call an entry point once eagerly on a
// device -- and on a stream whose eager launches will need tickets -- before capturing.
constexpr size_t kRingSlots = 16384, kChunkSlots = 262144;
struct TicketRange { size_t start, count; };
struct TicketPool {
    std::map<hipStream_t, std::pair<unsigned int *, size_t>> rings;     // stream -> (memory, next slot)
    unsigned int *chunk = nullptr;
    size_t chunk_next = 0;                                               // bump pointer behind the recycled ranges
    std::vector<TicketRange> free_ranges;                                // returned by released scopes, sorted by start
    std::map<int, std::vector<TicketRange>> scopes;                      // scope id -> ranges its captures hold
};
static std::mutex g_ticket_mu;
static std::map<int, TicketPool> g_ticket_pools;
static int g_next_scope = 1;
static thread_local int t_scope = 0;

static void free_range(TicketPool &p, TicketRange r)
{
    auto it = p.free_ranges.begin();
    while (it != p.free_ranges.end() && it->start < r.start) ++it;
    it = p.free_ranges.insert(it, r);
    if (it + 1 != p.free_ranges.end() && it->start + it->count == (it + 1)->start) {     // merge with the successor
        it->count += (it + 1)->count;
        p.free_ranges.erase(it + 1);
    }
    if (it != p.free_ranges.begin() && (it - 1)->start + (it - 1)->count == it->start) {   // ... and the predecessor
        (it - 1)->count += it->count;
        it = p.free_ranges.erase(it) - 1;
    }
    if (it->start + it->count == p.chunk_next) {                                            // the tail goes back to the bump pointer
        p.chunk_next = it->start;
        p.free_ranges.erase(it);
    }
}

int acquire_tickets(hipStream_t s, int n, unsigned int **ptr, int kind)
{
    CGIC_REQUIRE(n > 0 && (size_t)n <= kRingSlots / 4, CGIC_ERR_INVALID, "acquire_tickets: bad count %d", n);
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    CGIC_HIP_TRY(hipStreamIsCapturing(s, &cap));
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    TicketPool &p = g_ticket_pools[2 * dev + (kind ? 1 : 0)];      // (kind 1: the router's refinement headers -- their own memory, their own contract)
    if (cap == hipStreamCaptureStatusNone) {
        // The zero-fill has to be ORDERED in front of the first launch that uses the memory.  hipMemset() of device memory returns
        // before the fill has run and runs on the null stream, which the (non-blocking) streams of the callers do not wait for:
        // the first launch on a NEW stream could start ahead of its ring's fill and have its tickets wiped under it -- a lost
        // "last workgroup" (a loss that is never written, a slot left dirty for whoever gets it next) or, for kernels whose
        // workgroups wait for each other's flags, a launch that never ends.  Found in round 6 (tests/conftest.py checks after every
        // GPU test that the pools are all-zero: a tiled image's shape groups on fresh parallel streams left 285 dirty words; the
        // two fresh threads + streams of test_captured_ticket_slots_are_recycled... hung once in ~15 runs of the suite).
        // Ring: filled on the caller's own stream (stream order).  Capture pool: filled once, then the device is synchronised.
        if (!p.chunk) {
            CGIC_HIP_TRY(hipMalloc((void **)&p.chunk, sizeof(unsigned int) * kTicketStride * kChunkSlots));
            CGIC_HIP_TRY(hipMemset(p.chunk, 0, sizeof(unsigned int) * kTicketStride * kChunkSlots));
            CGIC_HIP_TRY(hipDeviceSynchronize());
        }
        auto &ring = p.rings[s];
        if (!ring.first) {
            CGIC_HIP_TRY(hipMalloc((void **)&ring.first, sizeof(unsigned int) * kTicketStride * kRingSlots));
            CGIC_HIP_TRY(hipMemsetAsync(ring.first, 0, sizeof(unsigned int) * kTicketStride * kRingSlots, s));
            ring.second = 0;
        }
        if (ring.second % kRingSlots + (size_t)n > kRingSlots) ring.second += kRingSlots - ring.second % kRingSlots;   // no wrap inside a range
        *ptr = ring.first + (ring.second % kRingSlots) * kTicketStride;
        ring.second += (size_t)n;
        return CGIC_OK;
    }
    CGIC_REQUIRE(p.chunk, CGIC_ERR_INVALID, "call once outside stream capture on this device before capturing");
    size_t start = kChunkSlots;
    for (auto it = p.free_ranges.begin(); it != p.free_ranges.end(); ++it)               // first fit among the recycled ranges
        if (it->count >= (size_t)n) {
            start = it->start;
            it->start += (size_t)n;
            it->count -= (size_t)n;
            if (!it->count) p.free_ranges.erase(it);
            break;
        }
    if (start == kChunkSlots) {
        CGIC_REQUIRE(p.chunk_next + (size_t)n <= kChunkSlots, CGIC_ERR_INVALID,
                     "the pool of ticket slots for captured launches is used up (%zu slots): release the scopes of destroyed graphs "
                     "(cgic_ticket_scope_release)", kChunkSlots);
        start = p.chunk_next;
        p.chunk_next += (size_t)n;
    }
    if (t_scope) p.scopes[t_scope].push_back(TicketRange{start, (size_t)n});
    *ptr = p.chunk + start * kTicketStride;
    return CGIC_OK;
}

int ensure_dynamic_lds(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> done;
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = done[std::make_pair(fn, dev)];
    if (bytes > have) {
        CGIC_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return CGIC_OK;
}

struct DevImage {
    int32_t *len = nullptr;
    uint32_t *code = nullptr;
    uint32_t *lut = nullptr;
    int32_t *child = nullptr;
};

struct Table {
    int n = 0, max_len = 0, words = 1, lut_bits = 1;
    std::vector<int32_t> len;     // [n]
    std::vector<uint32_t> code;   // [n * words], MSB first
    std::vector<uint32_t> lut;    // [1 << lut_bits]: (sym << 8) | len, or (node << 8) | 0 for long codes
    std::vector<int32_t> child;   // [2 * nodes]: >= 0 node id, < 0 = ~symbol
    std::mutex mu;
    std::map<int, DevImage> dev;
};

// ---- CPython heapq on node ids, ordered by freq only --------------------------
class PyHeap {
  public:
    explicit PyHeap(const std::vector<int64_t> &freq) : f_(freq) {}
    size_t size() const { return a_.size(); }
    void push(int id)
    {
        a_.push_back(id);
        bubble_up(0, a_.size() - 1);
    }
    int pop()
    {
        int last = a_.back();
        a_.pop_back();
        if (a_.empty()) return last;
        int top = a_[0];
        a_[0] = last;
        // walk the hole to a leaf, promoting the smaller child (right unless left < right)
        size_t pos = 0, end = a_.size(), child = 1;
        int item = a_[0];
        while (child < end) {
            size_t right = child + 1;
            if (right < end && !(f_[a_[child]] < f_[a_[right]])) child = right;
            a_[pos] = a_[child];
            pos = child;
            child = 2 * pos + 1;
        }
        a_[pos] = item;
        bubble_up(0, pos);
        return top;
    }

  private:
    void bubble_up(size_t start, size_t pos)
    {
        int item = a_[pos];
        while (pos > start) {
            size_t parent = (pos - 1) >> 1;
            if (f_[item] < f_[a_[parent]]) {
                a_[pos] = a_[parent];
                pos = parent;
            } else
                break;
        }
        a_[pos] = item;
    }
    const std::vector<int64_t> &f_;
    std::vector<int> a_;
};

static void finish_table(Table *t)
{
    // decode trie from the codes (full binary tree for n >= 2)
    t->child.assign(2, INT32_MIN);
    int nodes = 1;
    for (int s = 0; s < t->n; ++s) {
        int cur = 0;
        for (int b = 0; b < t->len[s]; ++b) {
            int bit = (t->code[(size_t)s * t->words + b / 32] >> (31 - b % 32)) & 1;
            if (b == t->len[s] - 1) {
                t->child[2 * cur + bit] = ~s;
            } else {
                int nx = t->child[2 * cur + bit];
                if (nx == INT32_MIN) {
                    nx = nodes++;
                    t->child.resize(2 * (size_t)nodes, INT32_MIN);
                    t->child[2 * cur + bit] = nx;
                }
                cur = nx;
            }
        }
    }
    // LUT on the first lut_bits bits
    // 13 bits: the whole of a Zipf-like 1024-symbol table (max_len 13) resolves with ONE LDS read per position;
    // with 12 bits, 5 % of the bit positions of a stream walked the trie, and a wave waits for its slowest lane
    t->lut_bits = t->max_len < kLutBitsMax ? (t->max_len < 1 ? 1 : t->max_len) : kLutBitsMax;
    const int LB = t->lut_bits;
    t->lut.assign((size_t)1 << LB, 0);
    for (uint32_t w = 0; w < (1u << LB); ++w) {
        int cur = 0, depth = 0;
        uint32_t entry = 0;
        for (;;) {
            if (depth == LB) { entry = ((uint32_t)cur << 8) | 0u; break; }   // long code: continue from `cur`
            int bit = (w >> (LB - 1 - depth)) & 1;
            int nx = t->child[2 * cur + bit];
            ++depth;
            if (nx == INT32_MIN) { entry = 0xFFFFFF00u; break; }             // no such prefix
            if (nx < 0) { entry = ((uint32_t)(~nx) << 8) | (uint32_t)depth; break; }
            cur = nx;
        }
        t->lut[w] = entry;
    }
}

static int upload(Table *t, int device, DevImage *img)
{
    CGIC_HIP_TRY(hipMalloc((void **)&img->len, sizeof(int32_t) * t->len.size()));
    CGIC_HIP_TRY(hipMalloc((void **)&img->code, sizeof(uint32_t) * t->code.size()));
    CGIC_HIP_TRY(hipMalloc((void **)&img->lut, sizeof(uint32_t) * t->lut.size()));
    CGIC_HIP_TRY(hipMalloc((void **)&img->child, sizeof(int32_t) * t->child.size()));
    CGIC_HIP_TRY(hipMemcpy(img->len, t->len.data(), sizeof(int32_t) * t->len.size(), hipMemcpyHostToDevice));
    CGIC_HIP_TRY(hipMemcpy(img->code, t->code.data(), sizeof(uint32_t) * t->code.size(), hipMemcpyHostToDevice));
    CGIC_HIP_TRY(hipMemcpy(img->lut, t->lut.data(), sizeof(uint32_t) * t->lut.size(), hipMemcpyHostToDevice));
    CGIC_HIP_TRY(hipMemcpy(img->child, t->child.data(), sizeof(int32_t) * t->child.size(), hipMemcpyHostToDevice));
    (void)device;
    return CGIC_OK;
}

int table_device_view(const cgic_table *ct, TableDev *out)
{
    Table *t = const_cast<Table *>(reinterpret_cast<const Table *>(ct));
    CGIC_REQUIRE(t, CGIC_ERR_INVALID, "table is NULL");
    int device = 0;
    CGIC_HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(t->mu);
    auto it = t->dev.find(device);
    if (it == t->dev.end()) {
        DevImage img;
        int rc = upload(t, device, &img);   // synchronous, once per (table, device)
        if (rc) return rc;
        it = t->dev.emplace(device, img).first;
    }
    out->len = it->second.len;
    out->code = it->second.code;
    out->lut = it->second.lut;
    out->child = it->second.child;
    out->n = t->n;
    out->words = t->words;
    out->max_len = t->max_len;
    out->lut_bits = t->lut_bits;
    out->n_nodes = (int)(t->child.size() / 2);
    int min_len = t->max_len;
    for (int v : t->len) if (v > 0 && v < min_len) min_len = v;
    if (min_len < 1) min_len = 1;
    const int hops = (kWave + min_len - 1) / min_len;     // codewords that can start inside one chunk
    int r = 0;
    while ((1 << r) < hops) ++r;
    out->dbl_rounds = r;
    int g = 0;                                             // gcd of the code lengths (see TableDev::len_gcd)
    for (int v : t->len) {
        int a = v, b = g;
        while (b) { const int c = a % b; a = b; b = c; }
        if (v > 0) g = a;
    }
    out->len_gcd = g < 1 || g > 64 ? 1 : g;
    return CGIC_OK;
}

}  // namespace cgic

using namespace cgic;

extern "C" const char *cgic_last_error(void) { return g_err; }
extern "C" int cgic_abi_version(void) { return CGIC_ABI_VERSION; }

extern "C" int cgic_ticket_scope_begin(void)
{
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    CGIC_REQUIRE(t_scope == 0, CGIC_ERR_INVALID, "ticket_scope_begin: this thread already has an open scope (%d)", t_scope);
    t_scope = g_next_scope++;
    return t_scope;
}

extern "C" int cgic_ticket_scope_end(void)
{
    const int id = t_scope;
    CGIC_REQUIRE(id != 0, CGIC_ERR_INVALID, "ticket_scope_end: no open scope on this thread");
    t_scope = 0;
    return id;
}

extern "C" int cgic_ticket_scope_release(int scope)
{
    CGIC_REQUIRE(scope > 0, CGIC_ERR_INVALID, "ticket_scope_release: bad scope %d", scope);
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    int freed = 0;
    for (auto &kv : g_ticket_pools) {
        auto it = kv.second.scopes.find(scope);
        if (it == kv.second.scopes.end()) continue;
        for (const TicketRange &r : it->second) { free_range(kv.second, r); freed += (int)r.count; }
        kv.second.scopes.erase(it);
    }
    return freed;
}

// Test tool: how many 32-bit words of the ticket memory of the current device (kind 0: the ring of every stream that has one, and the
// used part of the pool for captured launches) are NOT zero once the device is idle.  Every user hands its slots back all-zero, so the
// answer is 0 whenever no launch is in flight; anything else is a slot a later launch will trip over (wrong bytes, or workgroups that
// wait forever).  tests/conftest.py checks it after every GPU test.  Synchronises the device.
// (dev) the first `cap` dirty words as (where, slot, word, value): where = -1 the capture pool, k >= 0 the k-th ring
extern "C" int cgic_ticket_pool_dirty_dump(unsigned int *out4, int cap)
{
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    CGIC_HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    auto it = g_ticket_pools.find(2 * dev);
    if (it == g_ticket_pools.end()) return 0;
    const TicketPool &p = it->second;
    int n = 0;
    std::vector<unsigned int> host;
    auto scan = [&](const unsigned int *devp, size_t slots, int where) -> int {
        if (!devp || !slots) return CGIC_OK;
        host.resize(slots * kTicketStride);
        CGIC_HIP_TRY(hipMemcpy(host.data(), devp, host.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < host.size() && n < cap; ++i)
            if (host[i]) { out4[4 * n] = (unsigned int)where; out4[4 * n + 1] = (unsigned int)(i / kTicketStride); out4[4 * n + 2] = (unsigned int)(i % kTicketStride); out4[4 * n + 3] = host[i]; ++n; }
        return CGIC_OK;
    };
    int rc = scan(p.chunk, p.chunk_next, -1);
    if (rc) return rc;
    int k = 0;
    for (const auto &kv : p.rings) { rc = scan(kv.second.first, kRingSlots, k++); if (rc) return rc; }
    return n;
}

extern "C" long long cgic_ticket_pool_dirty_words(void)
{
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    CGIC_HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    auto it = g_ticket_pools.find(2 * dev);
    if (it == g_ticket_pools.end()) return 0;
    const TicketPool &p = it->second;
    long long dirty = 0;
    std::vector<unsigned int> host;
    auto scan = [&](const unsigned int *devp, size_t slots) -> int {
        if (!devp || !slots) return CGIC_OK;
        host.resize(slots * kTicketStride);
        CGIC_HIP_TRY(hipMemcpy(host.data(), devp, host.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
        for (unsigned int v : host) dirty += v != 0u;
        return CGIC_OK;
    };
    int rc = scan(p.chunk, p.chunk_next);
    if (rc) return rc;
    for (const auto &kv : p.rings) { rc = scan(kv.second.first, kRingSlots); if (rc) return rc; }
    return dirty;
}

extern "C" int cgic_ticket_slots_in_use(void)
{
    int dev = 0;
    CGIC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    // both pools of the device (kind 0: tickets; kind 1: the router's refinement headers): cgic_ticket_scope_release frees across both
    size_t used = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const TicketPool &p = g_ticket_pools[2 * dev + kind];
        used += p.chunk_next;
        for (const TicketRange &r : p.free_ranges) used -= r.count;
    }
    return (int)used;
}

extern "C" int cgic_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return CGIC_ERR_HIP;
    }
    return n;
}

extern "C" int cgic_table_create(const int64_t *freq, const int32_t *order, int n, cgic_table **out)
{
    CGIC_REQUIRE(freq && out, CGIC_ERR_INVALID, "table_create: NULL argument");
    CGIC_REQUIRE(n >= 2 && n <= 65536, CGIC_ERR_INVALID, "table_create: n=%d out of range [2, 65536]", n);
    std::vector<char> seen((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        int s = order ? order[i] : i;
        CGIC_REQUIRE(s >= 0 && s < n && !seen[s], CGIC_ERR_INVALID, "table_create: order is not a permutation");
        seen[s] = 1;
        CGIC_REQUIRE(freq[s] >= 0 && freq[s] < ((int64_t)1 << 46), CGIC_ERR_INVALID,
                     "table_create: freq[%d]=%lld outside [0, 2^46)", s, (long long)freq[s]);
    }
    Table *t = new (std::nothrow) Table();
    CGIC_REQUIRE(t, CGIC_ERR_NOMEM, "table_create: out of memory");
    t->n = n;
    // nodes: leaves 0..n-1 in PUSH order (node id = push position), then merged nodes
    std::vector<int64_t> f;
    std::vector<int> sym, left, right;
    f.reserve(2 * (size_t)n); sym.reserve(2 * (size_t)n); left.reserve(2 * (size_t)n); right.reserve(2 * (size_t)n);
    PyHeap heap(f);
    for (int i = 0; i < n; ++i) {                      // make_heap, indices_coding.py:46-49
        int s = order ? order[i] : i;
        f.push_back(freq[s]); sym.push_back(s); left.push_back(-1); right.push_back(-1);
        heap.push(i);
    }
    while (heap.size() > 1) {                          // merge_nodes, :51-60
        int a = heap.pop();
        int b = heap.pop();
        f.push_back(f[a] + f[b]); sym.push_back(-1); left.push_back(a); right.push_back(b);
        heap.push((int)f.size() - 1);
    }
    const int root = heap.pop();                       // make_codes, :73-75
    // depth-first walk, '0' = left, '1' = right (:62-71); codes as bit vectors
    t->len.assign((size_t)n, 0);
    std::vector<std::vector<bool>> bits((size_t)n);
    std::vector<std::pair<int, std::vector<bool>>> stack;
    stack.emplace_back(root, std::vector<bool>());
    while (!stack.empty()) {
        auto cur = std::move(stack.back());
        stack.pop_back();
        const int node = cur.first;
        if (sym[node] >= 0) {
            t->len[sym[node]] = (int32_t)cur.second.size();
            bits[sym[node]] = cur.second;
        } else {
            auto r = cur.second; r.push_back(true);
            auto l = std::move(cur.second); l.push_back(false);
            stack.emplace_back(right[node], std::move(r));
            stack.emplace_back(left[node], std::move(l));
        }
    }
    t->max_len = 0;
    for (int s = 0; s < n; ++s) if (t->len[s] > t->max_len) t->max_len = t->len[s];
    t->words = t->max_len > 0 ? (t->max_len + 31) / 32 : 1;
    t->code.assign((size_t)n * t->words, 0u);
    for (int s = 0; s < n; ++s)
        for (size_t b = 0; b < bits[s].size(); ++b)
            if (bits[s][b]) t->code[(size_t)s * t->words + b / 32] |= 1u << (31 - b % 32);
    finish_table(t);
    *out = reinterpret_cast<cgic_table *>(t);
    return CGIC_OK;
}

extern "C" int cgic_table_binary(cgic_table **out)
{
    CGIC_REQUIRE(out, CGIC_ERR_INVALID, "table_binary: NULL argument");
    Table *t = new (std::nothrow) Table();
    CGIC_REQUIRE(t, CGIC_ERR_NOMEM, "table_binary: out of memory");
    t->n = 2; t->max_len = 1; t->words = 1;            // mask_coding.py:11-12
    t->len = {1, 1};
    t->code = {0x00000000u, 0x80000000u};
    finish_table(t);
    *out = reinterpret_cast<cgic_table *>(t);
    return CGIC_OK;
}

extern "C" void cgic_table_destroy(cgic_table *ct)
{
    Table *t = reinterpret_cast<Table *>(ct);
    if (!t) return;
    for (auto &kv : t->dev) {
        (void)hipFree(kv.second.len); (void)hipFree(kv.second.code);
        (void)hipFree(kv.second.lut); (void)hipFree(kv.second.child);
    }
    delete t;
}

extern "C" int cgic_table_num_symbols(const cgic_table *t) { return t ? reinterpret_cast<const Table *>(t)->n : CGIC_ERR_INVALID; }
extern "C" int cgic_table_max_len(const cgic_table *t) { return t ? reinterpret_cast<const Table *>(t)->max_len : CGIC_ERR_INVALID; }
extern "C" int cgic_table_words(const cgic_table *t) { return t ? reinterpret_cast<const Table *>(t)->words : CGIC_ERR_INVALID; }

extern "C" int cgic_table_get(const cgic_table *ct, int32_t *len, uint32_t *code)
{
    const Table *t = reinterpret_cast<const Table *>(ct);
    CGIC_REQUIRE(t, CGIC_ERR_INVALID, "table_get: NULL table");
    if (len) memcpy(len, t->len.data(), sizeof(int32_t) * t->len.size());
    if (code) memcpy(code, t->code.data(), sizeof(uint32_t) * t->code.size());
    return CGIC_OK;
}
