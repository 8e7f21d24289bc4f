/* Typed CPython binding for the hot entry points of libsdnq_hip.so (include/sdnq_hip.h).
 *
 * An eager (no hipGraph) diffusion step makes ~900 calls into the library and is bound by the host: a ctypes call with 15-20
 * arguments costs ~10 us of argument conversion, the launch it wraps ~5 us.  This module exposes the SAME named entry points --
 * resolved here, by name, from the library whose path init() is given; nothing is re-implemented and no caller-supplied code address
 * is ever called -- through METH_FASTCALL wrappers that convert Python ints / None with the C API (~0.5 us per call).  Everything that
 * is not listed below stays on ctypes (sdnq_amd/_lib.py). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <dlfcn.h>
#include <stdint.h>

#include "../../include/sdnq_hip.h"

typedef long long i64;
typedef void* vp;

/* args[0..n) -> 64-bit integers (None = 0 = NULL); exactly `want` of them */
static int ints(PyObject* const* args, Py_ssize_t n, Py_ssize_t want, i64* a, const char* name) {
    if (n != want) { PyErr_Format(PyExc_TypeError, "%s takes %zd arguments (%zd given)", name, want, n); return -1; }
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (args[i] == Py_None) { a[i] = 0; continue; }
        a[i] = PyLong_AsLongLong(args[i]);
        if (a[i] == -1 && PyErr_Occurred()) return -1;
    }
    return 0;
}

#define P(i) ((vp)(uintptr_t)a[i])
#define I(i) ((int)a[i])
#define L(i) ((int64_t)a[i])

static __typeof__(&sdnq_hip_rowquant) f_rowquant;
static __typeof__(&sdnq_hip_scaled_mm) f_scaled_mm;
static __typeof__(&sdnq_hip_linear_w8a8) f_linear_w8a8;
static __typeof__(&sdnq_hip_scaled_mm_grouped) f_scaled_mm_grouped;
static __typeof__(&sdnq_hip_scaled_mm_lowrank) f_scaled_mm_lowrank;
static __typeof__(&sdnq_hip_lowrank_down) f_lowrank_down;
static __typeof__(&sdnq_hip_linear_w8a16) f_linear_w8a16;
static __typeof__(&sdnq_hip_linear_w8a16_grouped) f_linear_w8a16_grouped;
static __typeof__(&sdnq_hip_linear_float) f_linear_float;
static __typeof__(&sdnq_hip_prefetch_hint) f_prefetch_hint;
static __typeof__(&sdnq_hip_linear_w8a8_fused) f_linear_w8a8_fused;

#define NOT_READY(f) if (!(f)) { PyErr_SetString(PyExc_RuntimeError, "sdnq_amd._binding.init(path) has not been called"); return NULL; }

static PyObject* w_rowquant(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[15]; (void)s; NOT_READY(f_rowquant);
    if (ints(args, n, 15, a, "sdnq_hip_rowquant")) return NULL;
    return PyLong_FromLong(f_rowquant(P(0), I(1), L(2), L(3), L(4), I(5), I(6), P(7), (float*)P(8), (int32_t*)P(9), P(10), P(11), L(12), (float*)P(13), P(14)));
}
static PyObject* w_scaled_mm(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[15]; (void)s; NOT_READY(f_scaled_mm);
    if (ints(args, n, 15, a, "sdnq_hip_scaled_mm")) return NULL;
    return PyLong_FromLong(f_scaled_mm(I(0), P(1), P(2), (const float*)P(3), (const float*)P(4), P(5), I(6), I(7), L(8), P(9), I(10), L(11), L(12), L(13), P(14)));
}
static PyObject* w_linear_w8a8(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[17]; (void)s; NOT_READY(f_linear_w8a8);
    if (ints(args, n, 17, a, "sdnq_hip_linear_w8a8")) return NULL;
    return PyLong_FromLong(f_linear_w8a8(I(0), P(1), I(2), L(3), L(4), L(5), I(6), P(7), (float*)P(8), P(9), (const float*)P(10), P(11), I(12), P(13), I(14), L(15), P(16)));
}
static PyObject* w_scaled_mm_grouped(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[12]; (void)s; NOT_READY(f_scaled_mm_grouped);
    if (ints(args, n, 12, a, "sdnq_hip_scaled_mm_grouped")) return NULL;
    return PyLong_FromLong(f_scaled_mm_grouped(I(0), P(1), (const float*)P(2), (const SdnqGemmUnit*)P(3), L(4), L(5), I(6), P(7), I(8), L(9), L(10), P(11)));
}
static PyObject* w_scaled_mm_lowrank(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[21]; (void)s; NOT_READY(f_scaled_mm_lowrank);
    if (ints(args, n, 21, a, "sdnq_hip_scaled_mm_lowrank")) return NULL;
    return PyLong_FromLong(f_scaled_mm_lowrank(I(0), P(1), P(2), (const float*)P(3), (const float*)P(4), P(5), I(6), P(7), P(8), I(9), I(10), (const int32_t*)P(11),
                                               (const float*)P(12), (const float*)P(13), (const float*)P(14), P(15), I(16), L(17), L(18), L(19), P(20)));
}
static PyObject* w_lowrank_down(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[10]; (void)s; NOT_READY(f_lowrank_down);
    if (ints(args, n, 10, a, "sdnq_hip_lowrank_down")) return NULL;
    return PyLong_FromLong(f_lowrank_down(P(0), I(1), L(2), L(3), L(4), P(5), I(6), I(7), P(8), P(9)));
}
static PyObject* w_linear_w8a16(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[12]; (void)s; NOT_READY(f_linear_w8a16);
    if (ints(args, n, 12, a, "sdnq_hip_linear_w8a16")) return NULL;
    return PyLong_FromLong(f_linear_w8a16(P(0), I(1), P(2), (const float*)P(3), (const float*)P(4), P(5), P(6), L(7), L(8), L(9), L(10), P(11)));
}
static PyObject* w_linear_w8a16_grouped(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[11]; (void)s; NOT_READY(f_linear_w8a16_grouped);
    if (ints(args, n, 11, a, "sdnq_hip_linear_w8a16_grouped")) return NULL;
    return PyLong_FromLong(f_linear_w8a16_grouped(P(0), I(1), (const struct SdnqGemmUnit*)P(2), L(3), L(4), I(5), P(6), L(7), L(8), L(9), P(10)));
}
static PyObject* w_linear_float(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[10]; (void)s; NOT_READY(f_linear_float);
    if (ints(args, n, 10, a, "sdnq_hip_linear_float")) return NULL;
    return PyLong_FromLong(f_linear_float(P(0), P(1), P(2), I(3), P(4), L(5), L(6), L(7), L(8), P(9)));
}

static PyObject* w_linear_w8a8_fused(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[14]; (void)s; NOT_READY(f_linear_w8a8_fused);
    if (ints(args, n, 14, a, "sdnq_hip_linear_w8a8_fused")) return NULL;
    return PyLong_FromLong(f_linear_w8a8_fused(I(0), P(1), I(2), L(3), L(4), L(5), P(6), (const float*)P(7), P(8), I(9), P(10), I(11), L(12), P(13)));
}

static PyObject* w_prefetch_hint(PyObject* s, PyObject* const* args, Py_ssize_t n) {
    i64 a[8]; (void)s; NOT_READY(f_prefetch_hint);
    if (ints(args, n, 8, a, "sdnq_hip_prefetch_hint")) return NULL;
    return PyLong_FromLong(f_prefetch_hint(P(0), L(1), P(2), L(3), P(4), L(5), P(6), L(7)));
}

/* init(path): resolve the entry points from the library at `path` (the one sdnq_amd._lib loads with ctypes: same handle, RTLD_NOLOAD
 * is not required -- dlopen of an already loaded path returns it) */
static PyObject* w_init(PyObject* s, PyObject* arg) {
    (void)s;
    const char* path = PyUnicode_AsUTF8(arg);
    if (!path) return NULL;
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { PyErr_Format(PyExc_OSError, "dlopen(%s): %s", path, dlerror()); return NULL; }
#define R(name) do { *(void**)(&f_##name) = dlsym(h, "sdnq_hip_" #name); if (!f_##name) { PyErr_SetString(PyExc_OSError, "missing symbol sdnq_hip_" #name); return NULL; } } while (0)
    R(rowquant); R(scaled_mm); R(linear_w8a8); R(scaled_mm_grouped); R(scaled_mm_lowrank); R(lowrank_down); R(linear_w8a16);
    R(linear_w8a16_grouped); R(linear_float); R(prefetch_hint); R(linear_w8a8_fused);
#undef R
    Py_RETURN_NONE;
}

#define M(name) {"sdnq_hip_" #name, (PyCFunction)(void (*)(void))w_##name, METH_FASTCALL, "see include/sdnq_hip.h"}
static PyMethodDef methods[] = {M(rowquant), M(scaled_mm), M(linear_w8a8), M(scaled_mm_grouped), M(scaled_mm_lowrank), M(lowrank_down),
                                M(linear_w8a16), M(linear_w8a16_grouped), M(linear_float), M(prefetch_hint), M(linear_w8a8_fused),
                                {"init", (PyCFunction)w_init, METH_O, "init(path of libsdnq_hip.so)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_binding", "typed binding of libsdnq_hip.so's hot entry points", -1, methods,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__binding(void) { return PyModule_Create(&moddef); }
