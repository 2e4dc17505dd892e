// lslam_pyhost — host-side packing of a Python list of (n_i, d) NumPy arrays into ONE buffer, in C.
//
// The reference hands its scans from stage to stage as a Python list of ~5 000 small float64 arrays
// (modules/localization.py:115-124, modules/ogm.py:45-64).  Turning that list into the packed CSR buffer the
// C ABI takes costs the interpreter 6-10 ms (np.concatenate: one thread, one pass; plus a Python loop for the
// sizes) — several times the GPU time of the whole stage.  pack_list() walks the list through the buffer
// protocol (no Python-level loop), then copies every scan into the destination with a few threads, converting
// float32 <-> float64 on the way and checking finiteness in the same pass.  The destination is normally the
// library's pinned staging buffer (lslam_host_staging), so the upload that follows is one DMA at PCIe speed.
//
// pack_list(clouds: list, dst_addr: int, dst_capacity_points: int, dim: int, dst_is_f64: int, off_addr: int, threads: int)
//   -> (total_points, all_finite, float64_elements)   — all_finite = -1: the destination holds fewer than total_points, nothing was copied —
//   or None when an element is not a C-contiguous (n_i, dim) float32/float64 array (the caller then packs with NumPy).
//   off_addr: int64[len + 1] filled with the point offsets.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Item { const void* ptr; int64_t n; int is_f64; };

// copies and converts; finiteness is that of the CONVERTED value (a finite double beyond the float32 range becomes ±inf)
template <typename S, typename D>
bool copy_check(const S* s, D* d, int64_t count) {
  bool finite = true;
  for (int64_t i = 0; i < count; ++i) {
    const D v = (D)s[i];
    finite &= std::isfinite(v);
    d[i] = v;
  }
  return finite;
}

PyObject* pack_list(PyObject*, PyObject* args) {
  PyObject* list;
  unsigned long long dst_addr, off_addr;
  long long cap;
  int dim, dst_is_f64, threads;
  if (!PyArg_ParseTuple(args, "OKLiiKi", &list, &dst_addr, &cap, &dim, &dst_is_f64, &off_addr, &threads)) return nullptr;
  if (!PyList_Check(list) || dim < 1 || dim > 3) Py_RETURN_NONE;
  const Py_ssize_t K = PyList_GET_SIZE(list);
  std::vector<Item> items((size_t)K);
  // The buffer views are HELD until the copying threads have joined: the GIL is dropped while they read the arrays' memory, and
  // an exported buffer pins it (another Python thread that resizes an array or edits the list cannot free what is being read).
  std::vector<Py_buffer> views;
  views.reserve((size_t)K);
  auto release_views = [&]() { for (auto& v : views) PyBuffer_Release(&v); views.clear(); };
  int64_t* off = reinterpret_cast<int64_t*>(off_addr);
  int64_t total = 0, n_f64 = 0;
  off[0] = 0;
  for (Py_ssize_t k = 0; k < K; ++k) {
    Py_buffer b;
    if (PyObject_GetBuffer(PyList_GET_ITEM(list, k), &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); release_views(); Py_RETURN_NONE; }
    views.push_back(b);
    const bool f64 = b.format && b.format[0] == 'd' && b.format[1] == 0, f32 = b.format && b.format[0] == 'f' && b.format[1] == 0;
    const bool ok = (f64 || f32) && b.ndim == 2 && b.shape[1] == dim;
    if (!ok) { release_views(); Py_RETURN_NONE; }
    items[(size_t)k] = Item{b.buf, (int64_t)b.shape[0], f64 ? 1 : 0}; total += b.shape[0]; off[k + 1] = total; n_f64 += f64 ? 1 : 0;
  }
  if (total > cap) { release_views(); return Py_BuildValue("(LiL)", (long long)total, -1, (long long)n_f64); }      // destination too small: nothing copied, the caller grows it
  std::atomic<int> finite{1};
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if (total < 65536) threads = 1;
  Py_BEGIN_ALLOW_THREADS
  auto work = [&](int t) {
    // thread t copies the scans whose FIRST point lies in its share [lo, hi) of the points
    const int64_t lo = total * t / threads, hi = total * (t + 1) / threads;
    bool fin = true;
    for (Py_ssize_t k = 0; k < K; ++k) {
      if (off[k] < lo) continue;
      if (off[k] >= hi) break;
      const Item& it = items[(size_t)k];
      const int64_t cnt = it.n * dim, at = off[k] * dim;
      if (dst_is_f64) {
        double* d = reinterpret_cast<double*>(dst_addr) + at;
        fin &= it.is_f64 ? copy_check(static_cast<const double*>(it.ptr), d, cnt) : copy_check(static_cast<const float*>(it.ptr), d, cnt);
      } else {
        float* d = reinterpret_cast<float*>(dst_addr) + at;
        fin &= it.is_f64 ? copy_check(static_cast<const double*>(it.ptr), d, cnt) : copy_check(static_cast<const float*>(it.ptr), d, cnt);
      }
    }
    if (!fin) finite.store(0);
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  Py_END_ALLOW_THREADS
  release_views();
  return Py_BuildValue("(LiL)", (long long)total, finite.load(), (long long)n_f64);
}

PyMethodDef methods[] = {{"pack_list", pack_list, METH_VARARGS, "pack a list of (n_i, d) arrays into one buffer (see the file header)"},
                         {nullptr, nullptr, 0, nullptr}};
PyModuleDef module = {PyModuleDef_HEAD_INIT, "_lslam_pyhost", "host-side packing helpers of lidar_slam_amd", -1, methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__lslam_pyhost(void) { return PyModule_Create(&module); }
