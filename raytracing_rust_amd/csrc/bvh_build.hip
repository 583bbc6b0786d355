// BVH construction on the GPU: the order in which BvhNode::from_slice (src/ray_tracing/acceleration/bvh.rs:26-71)
// arranges the objects of a list, computed level by level for all nodes of a level at once.
//
// The reference builds its tree recursively: bounding box of the node's objects, longest axis of that box
// (aabb.rs:77-89), stable sort of the objects by the lower bound of their boxes on that axis (bvh.rs:61-62,75-83),
// split in the middle (bvh.rs:64-66); two objects are ordered by one comparison (bvh.rs:48-57: the second one comes
// first unless the first compares Less), one object is its node's left and right child.  The SHAPE of the tree
// depends on the number of objects only; what the sorts decide is which object ends up where.  So the whole build
// is a permutation: this file computes it, the host layer hangs the tree on it (scene_graph.cpp).
//
// Per level: (1) every node's box from its objects' boxes (min/max of order-preserving integer images of the doubles,
// reduced within a wave first, then one atomic per wave and bound); (2) every node's axis; (3) one key per object:
// the lower bound on its node's axis; (4) a stable sort of every node's range, as two stable radix sorts over the
// whole array, LSD fashion: by key, then by the node's first position (an object outside every node of the level
// keeps a key of its own, its position, so it stays where it is); (5) the two-object nodes by their own rule.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "device_internal.h"

namespace {

// order-preserving image of a double in the unsigned integers (-0 and +0 share one: they compare equal)
__host__ __device__ inline unsigned long long enc(double x) {
  if (x == 0.0) x = 0.0;
  unsigned long long u;
  memcpy(&u, &x, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double dec(unsigned long long u) {
  u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
  double x;
  memcpy(&x, &u, 8);
  return x;
}

struct Seg { uint32_t start, end; };

// node of this level that holds position i, or -1 (segments are disjoint and sorted by start)
__device__ inline int find_seg(const Seg* segs, uint32_t n_segs, uint32_t i) {
  uint32_t lo = 0, hi = n_segs;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (segs[mid].start <= i) lo = mid + 1; else hi = mid; }
  if (lo == 0) return -1;
  return i < segs[lo - 1].end ? (int)(lo - 1) : -1;
}

__global__ void k_iota(uint32_t* a, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void k_box_init(unsigned long long* segbox, uint32_t n_segs) {  // lo[3] = +max, hi[3] = 0: identities of min / max
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_segs * 6u) segbox[i] = (i % 6u) < 3u ? ~0ull : 0ull;
}
// (1) BvhNode::from_slice, bvh.rs:28-33: the node's box is the merge of its objects' boxes
__global__ __launch_bounds__(256) void k_seg_bbox(const Seg* segs, uint32_t n_segs, const uint32_t* order, const double* boxes, uint32_t n,
                                                  int* seg_of, unsigned long long* segbox) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = i < n ? find_seg(segs, n_segs, i) : -1;
  if (i < n) seg_of[i] = j;
  unsigned long long v[6];
  for (int k = 0; k < 6; ++k) v[k] = k < 3 ? ~0ull : 0ull;
  if (j >= 0) { const double* b = boxes + 6 * (size_t)order[i]; for (int k = 0; k < 6; ++k) v[k] = enc(b[k]); }
  // a wave whose lanes all stand in one node reduces first: near the root every lane of the launch shares a node
  const int j0 = __shfl(j, 0);
  if (__all(j == j0)) {
    if (j0 < 0) return;
    for (int off = 32; off > 0; off >>= 1)
      for (int k = 0; k < 6; ++k) { const unsigned long long o = __shfl_down(v[k], off); v[k] = k < 3 ? (o < v[k] ? o : v[k]) : (o > v[k] ? o : v[k]); }
    if ((threadIdx.x & 63u) == 0)
      for (int k = 0; k < 6; ++k) { if (k < 3) atomicMin(&segbox[6 * (size_t)j0 + k], v[k]); else atomicMax(&segbox[6 * (size_t)j0 + k], v[k]); }
  } else if (j >= 0) {
    for (int k = 0; k < 6; ++k) { if (k < 3) atomicMin(&segbox[6 * (size_t)j + k], v[k]); else atomicMax(&segbox[6 * (size_t)j + k], v[k]); }
  }
}
// (2) Aabb::longest_axis, aabb.rs:77-89 (ties prefer x, then y)
__global__ void k_seg_axis(const unsigned long long* segbox, uint32_t n_segs, int* axis) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_segs) return;
  const unsigned long long* b = segbox + 6 * (size_t)j;
  const double xs = dec(b[3]) - dec(b[0]), ys = dec(b[4]) - dec(b[1]), zs = dec(b[5]) - dec(b[2]);
  axis[j] = (xs >= ys && xs >= zs) ? 0 : (ys >= zs ? 1 : 2);
}
// (3) box_compare's operand (bvh.rs:75-83) as a sortable key, and the key of the second sort.  Nodes of two objects
// are not sorted (k_pairs orders them): their objects keep their positions like the ones outside every node.
__global__ __launch_bounds__(256) void k_keys(const Seg* segs, const int* seg_of, const int* axis, const uint32_t* order, const double* boxes,
                                              uint32_t n, unsigned long long* key, uint32_t* place, uint32_t* pos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = seg_of[i];
  pos[i] = i;
  if (j >= 0 && segs[j].end - segs[j].start >= 3u) { key[i] = enc(boxes[6 * (size_t)order[i] + axis[j]]); place[i] = segs[j].start; }
  else { key[i] = 0ull; place[i] = i; }
}
__global__ void k_gather(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// (5) bvh.rs:48-57: of two objects the first stays first only if it compares Less
__global__ void k_pairs(const Seg* segs, uint32_t n_segs, const int* axis, const double* boxes, uint32_t* order) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_segs || segs[j].end - segs[j].start != 2u) return;
  const uint32_t s = segs[j].start, a = order[s], b = order[s + 1];
  if (!(boxes[6 * (size_t)a + axis[j]] < boxes[6 * (size_t)b + axis[j]])) { order[s] = b; order[s + 1] = a; }
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T* as() const { return (T*)p; }
};

}  // namespace

#define BVH_TRY(call)                                                                                                   \
  do {                                                                                                                  \
    hipError_t e_ = (call);                                                                                             \
    if (e_ != hipSuccess) return rtx_internal_fail(ctx, RTX_ERR_HIP, (std::string("rtx_bvh_leaf_order: ") + #call + ": " + hipGetErrorString(e_)).c_str()); \
  } while (0)

extern "C" rtx_status rtx_bvh_leaf_order(rtx_context* ctx, uint32_t n, const double* boxes, uint32_t* order_out) {
  if (!ctx || (n && (!boxes || !order_out))) return rtx_internal_fail(ctx, RTX_ERR_INVALID, "rtx_bvh_leaf_order: null argument");
  if (n == 0) return RTX_OK;
  if (n >= (1u << 30)) return rtx_internal_fail(ctx, RTX_ERR_UNSUPPORTED, "rtx_bvh_leaf_order: more than 2^30 objects");
  BVH_TRY(hipSetDevice(rtx_internal_device(ctx)));
  // the nodes of every level (those with two objects or more), from the number of objects alone
  std::vector<std::vector<Seg>> levels;
  {
    std::vector<Seg> cur;
    if (n >= 2) cur.push_back(Seg{0, n});
    while (!cur.empty()) {
      levels.push_back(cur);
      std::vector<Seg> next;
      for (const Seg& s : cur) {
        const uint32_t span = s.end - s.start;
        if (span < 3) continue;
        const uint32_t mid = s.start + span / 2;                     // bvh.rs:64
        if (mid - s.start >= 2) next.push_back(Seg{s.start, mid});
        if (s.end - mid >= 2) next.push_back(Seg{mid, s.end});
      }
      cur.swap(next);
    }
  }
  size_t max_segs = 1;
  for (const auto& l : levels) max_segs = std::max(max_segs, l.size());
  DevBuf d_boxes, d_order[2], d_key[2], d_place[2], d_pos[2], d_segs, d_segbox, d_axis, d_seg_of, d_tmp;
  BVH_TRY(d_boxes.alloc((size_t)n * 6 * sizeof(double)));
  for (int k = 0; k < 2; ++k) {
    BVH_TRY(d_order[k].alloc((size_t)n * 4)); BVH_TRY(d_key[k].alloc((size_t)n * 8));
    BVH_TRY(d_place[k].alloc((size_t)n * 4)); BVH_TRY(d_pos[k].alloc((size_t)n * 4));
  }
  BVH_TRY(d_segs.alloc(max_segs * sizeof(Seg)));
  BVH_TRY(d_segbox.alloc(max_segs * 6 * 8));
  BVH_TRY(d_axis.alloc(max_segs * 4));
  BVH_TRY(d_seg_of.alloc((size_t)n * 4));
  size_t tmp_a = 0, tmp_b = 0;
  BVH_TRY(rocprim::radix_sort_pairs(nullptr, tmp_a, d_key[0].as<unsigned long long>(), d_key[1].as<unsigned long long>(),
                                    d_pos[0].as<uint32_t>(), d_pos[1].as<uint32_t>(), n, 0, 64, 0));
  BVH_TRY(rocprim::radix_sort_pairs(nullptr, tmp_b, d_place[0].as<uint32_t>(), d_place[1].as<uint32_t>(),
                                    d_pos[0].as<uint32_t>(), d_pos[1].as<uint32_t>(), n, 0, 32, 0));
  BVH_TRY(d_tmp.alloc(std::max(tmp_a, tmp_b)));
  BVH_TRY(hipMemcpy(d_boxes.p, boxes, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
  const unsigned nb = (n + 255u) / 256u;
  int cur = 0;
  hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, 0, d_order[cur].as<uint32_t>(), n);
  for (const auto& segs : levels) {
    const uint32_t ns = (uint32_t)segs.size();
    const unsigned sb = (ns + 255u) / 256u;
    bool any_sort = false, any_pair = false;
    for (const Seg& s : segs) { if (s.end - s.start >= 3) any_sort = true; else any_pair = true; }
    BVH_TRY(hipMemcpyAsync(d_segs.p, segs.data(), ns * sizeof(Seg), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(k_box_init, dim3((ns * 6u + 255u) / 256u), dim3(256), 0, 0, d_segbox.as<unsigned long long>(), ns);
    hipLaunchKernelGGL(k_seg_bbox, dim3(nb), dim3(256), 0, 0, d_segs.as<Seg>(), ns, d_order[cur].as<uint32_t>(), d_boxes.as<double>(), n,
                       d_seg_of.as<int>(), d_segbox.as<unsigned long long>());
    hipLaunchKernelGGL(k_seg_axis, dim3(sb), dim3(256), 0, 0, d_segbox.as<unsigned long long>(), ns, d_axis.as<int>());
    if (any_sort) {
      hipLaunchKernelGGL(k_keys, dim3(nb), dim3(256), 0, 0, d_segs.as<Seg>(), d_seg_of.as<int>(), d_axis.as<int>(), d_order[cur].as<uint32_t>(),
                         d_boxes.as<double>(), n, d_key[0].as<unsigned long long>(), d_place[0].as<uint32_t>(), d_pos[0].as<uint32_t>());
      // stable sort by key, carrying the old positions; then by the node's first position, carrying them on
      size_t t = tmp_a;
      BVH_TRY(rocprim::radix_sort_pairs(d_tmp.p, t, d_key[0].as<unsigned long long>(), d_key[1].as<unsigned long long>(),
                                        d_pos[0].as<uint32_t>(), d_pos[1].as<uint32_t>(), n, 0, 64, 0));
      hipLaunchKernelGGL(k_gather, dim3(nb), dim3(256), 0, 0, d_place[0].as<uint32_t>(), d_pos[1].as<uint32_t>(), n, d_place[1].as<uint32_t>());
      t = tmp_b;
      BVH_TRY(rocprim::radix_sort_pairs(d_tmp.p, t, d_place[1].as<uint32_t>(), d_place[0].as<uint32_t>(),
                                        d_pos[1].as<uint32_t>(), d_pos[0].as<uint32_t>(), n, 0, 32, 0));
      hipLaunchKernelGGL(k_gather, dim3(nb), dim3(256), 0, 0, d_order[cur].as<uint32_t>(), d_pos[0].as<uint32_t>(), n, d_order[cur ^ 1].as<uint32_t>());
      cur ^= 1;
    }
    if (any_pair)
      hipLaunchKernelGGL(k_pairs, dim3(sb), dim3(256), 0, 0, d_segs.as<Seg>(), ns, d_axis.as<int>(), d_boxes.as<double>(), d_order[cur].as<uint32_t>());
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipStreamSynchronize(0));   // the level's segment list is re-used by the next upload
  }
  BVH_TRY(hipMemcpy(order_out, d_order[cur].p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return RTX_OK;
}
