/* Host-side fast path of the eager SDNQ Linear forward (sdnq_amd/_fastpath.so).
 *
 * An eager (no hipGraph) SDXL step makes 741 Linear calls and is bound by the HOST: 11.8 ms of Python per step around 7.3 ms of GPU
 * work (tools/eager_split.py, profiles/r06_eager_fastpath.txt) -- module call, grad-mode context, parameter identity checks, route
 * decisions, tensor keys, allocations, argument marshalling: ~16 us per layer call of which the launches themselves are 4-7.  This
 * module carries the three routes that make up 88 % of those calls through ONE C++ call each:
 *
 *   Plan(mod, input) -> Tensor | None | False
 *     * a member of a projection group picks up the output the group's launch already computed for it (ProjectionGroup._claim),
 *     * a layer whose input is its own runs the one-launch w8a8 Linear (sdnq_hip_linear_w8a8_fused) or the row quantizer + GEMM with the
 *       quantized activation in the stream's scratch buffer (sdnq_hip_linear_w8a8),
 *     None: this call is not one of those (few rows, a stream capture on the scratch route, another device, ...): the Python forward
 *     runs;  False: the plan is STALE (a parameter object, its storage or its version changed; a module-level switch of linear.py was
 *     flipped): the caller drops it and the Python forward builds a new one.
 *
 * Nothing here computes: the launches are the named entry points of libsdnq_hip.so (resolved by name from the library init() is
 * given, like binding.c), every decision restates sdnq_amd/linear.py (the line it mirrors is cited), and linear.py remains the complete
 * implementation -- with SDNQ_HIP_FAST_PLANS=0 or without this module it runs alone, bit-identical (tests/test_fastpath.py).
 * The weight-prefetch chain (linear._PrefetchChain) and the claim state of a projection group live here so that the Python path and
 * the plans see ONE copy of each.  The GIL is held throughout; per-thread state is thread_local.
 *
 * Reference seam: forward_func(self, input) (layers/__init__.py:29-30, quantized_linear_forward_int8_matmul linear_int8.py:101-107). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>

#include <ATen/ATen.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

#include <dlfcn.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/sdnq_hip.h"

namespace {

decltype(&sdnq_hip_linear_w8a8) f_linear_w8a8 = nullptr;
decltype(&sdnq_hip_linear_w8a8_fused) f_fused = nullptr;
decltype(&sdnq_hip_linear_w8a8_fused_supported) f_fused_supported = nullptr;
decltype(&sdnq_hip_prefetch_hint) f_hint = nullptr;
decltype(&sdnq_hip_stream_capture_id) f_capture_id = nullptr;

std::atomic<long> g_epoch{0};        // bumped whenever a module-level switch of linear.py is assigned: every plan made before is stale
std::atomic<long> g_fused_calls{0};  // calls that took the one-launch route (ops.fused_calls counts the Python path's)
std::atomic<long> g_plan_calls{0};   // calls a plan carried (tests / bench: proof that the fast path ran)
thread_local int t_no_reuse = 0;     // linear._ts.no_reuse (identity_reuse_disabled): no identity-keyed reuse on this thread

int float_code(at::ScalarType t) {  // ops.float_code
    if (t == at::kBFloat16) return SDNQ_BF16;
    if (t == at::kHalf) return SDNQ_F16;
    if (t == at::kFloat) return SDNQ_F32;
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the weight-prefetch chain (linear._LaunchUnit / _PrefetchChain.launch)
// ---------------------------------------------------------------------------------------------------------------------------
struct Unit {
    const void* p[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t b[4] = {0, 0, 0, 0};
    int nr = 0;
    int device = -1;
    std::weak_ptr<Unit> next;
};
thread_local std::weak_ptr<Unit> t_prev;
thread_local int64_t t_last_hint[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0]: hints handed over by this thread; then the last one (tests)

void unit_launch(const std::shared_ptr<Unit>& u) {
    auto prev = t_prev.lock();
    if (prev && prev != u) {
        auto nx = prev->next.lock();
        if (nx != u) prev->next = u;
    }
    t_prev = u;
    auto n1 = u->next.lock();
    if (!n1 || n1->device != u->device) return;
    auto n2 = n1->next.lock();
    const void* p[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t b[4] = {0, 0, 0, 0};
    int c = 0;
    for (int i = 0; i < n1->nr && c < 4; ++i, ++c) { p[c] = n1->p[i]; b[c] = n1->b[i]; }
    if (n2 && n2 != u && n2->device == u->device)
        for (int i = 0; i < n2->nr && c < 4; ++i, ++c) { p[c] = n2->p[i]; b[c] = n2->b[i]; }
    if (c == 0 || !f_hint) return;
    t_last_hint[0] += 1;
    for (int i = 0; i < 4; ++i) { t_last_hint[1 + 2 * i] = (int64_t)(uintptr_t)p[i]; t_last_hint[2 + 2 * i] = b[i]; }
    f_hint(p[0], b[0], p[1], b[1], p[2], b[2], p[3], b[3]);
}

struct UnitObj {
    PyObject_HEAD
    std::shared_ptr<Unit>* u;
};

PyObject* unit_new(PyTypeObject* type, PyObject* args, PyObject*) {
    PyObject* ranges;
    int device;
    if (!PyArg_ParseTuple(args, "Oi", &ranges, &device)) return nullptr;
    PyObject* seq = PySequence_Fast(ranges, "ranges must be a sequence of (address, bytes)");
    if (!seq) return nullptr;
    auto u = std::make_shared<Unit>();
    u->device = device;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    if (n > 4) { Py_DECREF(seq); PyErr_SetString(PyExc_ValueError, "at most four ranges"); return nullptr; }
    for (Py_ssize_t i = 0; i < n; ++i) {
        unsigned long long ptr;
        long long bytes;
        if (!PyArg_ParseTuple(PySequence_Fast_GET_ITEM(seq, i), "KL", &ptr, &bytes)) { Py_DECREF(seq); return nullptr; }
        u->p[i] = (const void*)(uintptr_t)ptr;
        u->b[i] = bytes;
    }
    u->nr = (int)n;
    Py_DECREF(seq);
    UnitObj* self = (UnitObj*)type->tp_alloc(type, 0);
    if (!self) return nullptr;
    self->u = new std::shared_ptr<Unit>(std::move(u));
    return (PyObject*)self;
}
void unit_dealloc(UnitObj* self) {
    delete self->u;
    Py_TYPE(self)->tp_free((PyObject*)self);
}
PyObject* unit_launch_py(UnitObj* self, PyObject*) {
    unit_launch(*self->u);
    Py_RETURN_NONE;
}
PyMethodDef unit_methods[] = {{"launch", (PyCFunction)unit_launch_py, METH_NOARGS, "this unit launches now: link it behind the thread's previous unit, hand its successors' weights to the library"},
                              {nullptr, nullptr, 0, nullptr}};
PyTypeObject UnitType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---------------------------------------------------------------------------------------------------------------------------
// claim state of a projection group (ProjectionGroup.last / _claim, linear.py)
// ---------------------------------------------------------------------------------------------------------------------------
struct TensorKey {  // linear.tensor_key
    const void* ptr = nullptr;
    int64_t offset = 0, version = 0;
    std::vector<int64_t> sizes, strides;
    bool matches(const at::Tensor& t) const {
        return t.data_ptr() == ptr && t.storage_offset() == offset && (int64_t)t._version() == version && t.sizes() == at::IntArrayRef(sizes) &&
               t.strides() == at::IntArrayRef(strides);
    }
    void take(const at::Tensor& t) {
        ptr = t.data_ptr();
        offset = t.storage_offset();
        version = (int64_t)t._version();
        sizes = t.sizes().vec();
        strides = t.strides().vec();
    }
};

void* current_stream_of(const at::Tensor& t) {
    if (!t.is_cuda()) return nullptr;
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

struct GroupObj {
    PyObject_HEAD
    int n;
    int wasted;          // consecutive computes whose outputs were not all claimed (ProjectionGroup._begin_compute)
    PyObject* input;     // strong; nullptr = nothing stored
    TensorKey* key;
    void* stream;
    std::vector<PyObject*>* outs;   // strong references, one [M, N] matrix per member
    std::vector<uint8_t>* pending;  // members that have not picked theirs up
    int n_pending;
};

void group_clear(GroupObj* g) {
    Py_CLEAR(g->input);
    for (PyObject* o : *g->outs) Py_XDECREF(o);
    g->outs->clear();
    std::fill(g->pending->begin(), g->pending->end(), (uint8_t)0);
    g->n_pending = 0;
}

PyObject* group_new(PyTypeObject* type, PyObject* args, PyObject*) {
    int n;
    if (!PyArg_ParseTuple(args, "i", &n)) return nullptr;
    if (n <= 0) { PyErr_SetString(PyExc_ValueError, "a group has at least one member"); return nullptr; }
    GroupObj* g = (GroupObj*)type->tp_alloc(type, 0);
    if (!g) return nullptr;
    g->n = n;
    g->wasted = 0;
    g->input = nullptr;
    g->key = new TensorKey();
    g->stream = nullptr;
    g->outs = new std::vector<PyObject*>();
    g->pending = new std::vector<uint8_t>((size_t)n, (uint8_t)0);
    g->n_pending = 0;
    return (PyObject*)g;
}
void group_dealloc(GroupObj* g) {
    group_clear(g);
    delete g->key;
    delete g->outs;
    delete g->pending;
    Py_TYPE(g)->tp_free((PyObject*)g);
}

// the stored output of member idx if `input` is the tensor the stored outputs were computed from (same object, unchanged, same stream);
// nullptr WITHOUT an exception: no claim
PyObject* group_claim(GroupObj* g, int idx, PyObject* input_obj, const at::Tensor& x) {
    if (g->input == nullptr || g->input != input_obj || idx < 0 || idx >= g->n || !(*g->pending)[(size_t)idx]) return nullptr;
    if (x.is_inference() || !g->key->matches(x) || current_stream_of(x) != g->stream) return nullptr;
    const at::Tensor& o = THPVariable_Unpack((*g->outs)[(size_t)idx]);
    std::vector<int64_t> shape = x.sizes().vec();
    shape.back() = o.size(-1);
    at::Tensor y = o.view(shape);  // .view(*input.shape[:-1], -1)
    (*g->pending)[(size_t)idx] = 0;
    if (--g->n_pending == 0) {  // every member has its output: hold on to nothing
        group_clear(g);
        g->wasted = 0;
    }
    return THPVariable_Wrap(y);
}

PyObject* group_claim_py(GroupObj* g, PyObject* args) {
    int idx;
    PyObject* input;
    if (!PyArg_ParseTuple(args, "iO", &idx, &input)) return nullptr;
    if (!THPVariable_Check(input)) Py_RETURN_NONE;
    try {
        PyObject* y = group_claim(g, idx, input, THPVariable_Unpack(input));
        if (y) return y;
        if (PyErr_Occurred()) return nullptr;
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    Py_RETURN_NONE;
}

PyObject* group_publish_py(GroupObj* g, PyObject* args) {
    PyObject *input, *outs;
    if (!PyArg_ParseTuple(args, "OO", &input, &outs)) return nullptr;
    if (!THPVariable_Check(input)) { PyErr_SetString(PyExc_TypeError, "input must be a tensor"); return nullptr; }
    PyObject* seq = PySequence_Fast(outs, "outs must be a sequence of tensors");
    if (!seq) return nullptr;
    if (PySequence_Fast_GET_SIZE(seq) != g->n) { Py_DECREF(seq); PyErr_SetString(PyExc_ValueError, "one output per member"); return nullptr; }
    for (Py_ssize_t i = 0; i < g->n; ++i)
        if (!THPVariable_Check(PySequence_Fast_GET_ITEM(seq, i))) { Py_DECREF(seq); PyErr_SetString(PyExc_TypeError, "outs must be tensors"); return nullptr; }
    try {
        const at::Tensor& x = THPVariable_Unpack(input);
        group_clear(g);
        if (x.is_inference()) { Py_DECREF(seq); Py_RETURN_NONE; }  // (no version counter: never served from a group)
        g->key->take(x);
        g->stream = current_stream_of(x);
    } catch (const std::exception& e) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    Py_INCREF(input);
    g->input = input;
    for (Py_ssize_t i = 0; i < g->n; ++i) {
        PyObject* o = PySequence_Fast_GET_ITEM(seq, i);
        Py_INCREF(o);
        g->outs->push_back(o);
        (*g->pending)[(size_t)i] = 1;
    }
    g->n_pending = g->n;
    Py_DECREF(seq);
    Py_RETURN_NONE;
}

PyObject* group_clear_py(GroupObj* g, PyObject*) {
    group_clear(g);
    Py_RETURN_NONE;
}
PyObject* group_pending_py(GroupObj* g, PyObject*) { return PyLong_FromLong(g->n_pending); }
// (input, outputs, indices not handed out yet) or None: what ProjectionGroup.last shows
PyObject* group_peek_py(GroupObj* g, PyObject*) {
    if (!g->input) Py_RETURN_NONE;
    PyObject* outs = PyList_New((Py_ssize_t)g->outs->size());
    if (!outs) return nullptr;
    for (size_t i = 0; i < g->outs->size(); ++i) {
        Py_INCREF((*g->outs)[i]);
        PyList_SET_ITEM(outs, (Py_ssize_t)i, (*g->outs)[i]);
    }
    PyObject* pend = PySet_New(nullptr);
    if (!pend) { Py_DECREF(outs); return nullptr; }
    for (int i = 0; i < g->n; ++i)
        if ((*g->pending)[(size_t)i]) {
            PyObject* v = PyLong_FromLong(i);
            PySet_Add(pend, v);
            Py_DECREF(v);
        }
    return Py_BuildValue("(ONN)", g->input, outs, pend);
}

PyMethodDef group_methods[] = {
    {"claim", (PyCFunction)group_claim_py, METH_VARARGS, "claim(idx, input) -> member idx's output as input.shape[:-1] + (N,), or None"},
    {"publish", (PyCFunction)group_publish_py, METH_VARARGS, "publish(input, outs): the group's launch computed outs from input"},
    {"clear", (PyCFunction)group_clear_py, METH_NOARGS, "forget the stored outputs"},
    {"pending", (PyCFunction)group_pending_py, METH_NOARGS, "how many stored outputs nobody has claimed"},
    {"peek", (PyCFunction)group_peek_py, METH_NOARGS, "(input, outputs, pending indices) or None"},
    {nullptr, nullptr, 0, nullptr}};
PyMemberDef group_members[] = {{"wasted", T_INT, offsetof(GroupObj, wasted), 0, "consecutive computes whose outputs were not all claimed"},
                               {nullptr, 0, 0, 0, nullptr}};
PyTypeObject GroupType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---------------------------------------------------------------------------------------------------------------------------
// per-(device, stream) scratch of the row quantizer (ops._workspace): grown on demand, the replaced buffer stays valid for the work
// already queued (the caching allocator does not hand a freed block to another stream)
// ---------------------------------------------------------------------------------------------------------------------------
struct WsKey {
    int dev;
    void* stream;
    bool operator==(const WsKey& o) const { return dev == o.dev && stream == o.stream; }
};
struct WsHash {
    size_t operator()(const WsKey& k) const { return std::hash<void*>()(k.stream) ^ (size_t)(k.dev * 0x9e3779b1u); }
};
std::unordered_map<WsKey, at::Tensor, WsHash>& workspaces() {
    static auto* m = new std::unordered_map<WsKey, at::Tensor, WsHash>();  // (never destroyed: outlives the HIP context teardown order)
    return *m;
}

uintptr_t workspace(const at::Tensor& like, void* stream, int64_t nbytes) {
    WsKey key{(int)like.device().index(), stream};
    auto& m = workspaces();
    auto it = m.find(key);
    if (it == m.end() || it->second.numel() < nbytes + 255) {
        at::Tensor buf = at::empty({std::max<int64_t>(nbytes, (int64_t)1 << 20) + 255}, like.options().dtype(at::kByte));
        m[key] = buf;
        return ((uintptr_t)buf.data_ptr() + 255) & ~(uintptr_t)255;
    }
    return ((uintptr_t)it->second.data_ptr() + 255) & ~(uintptr_t)255;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the plan
// ---------------------------------------------------------------------------------------------------------------------------
struct KeyEnt {
    PyObject* name;  // interned str, strong
    PyObject* ref;   // the object the plan was built from (strong; Py_None for an absent tensor)
    const void* ptr;
    int64_t version;
};

struct PlanObj {
    PyObject_HEAD
    vectorcallfunc vc;
    std::vector<KeyEnt>* keys;
    at::Tensor* wq;
    at::Tensor* ws;
    at::Tensor* bias;  // undefined tensor: none
    int mm, had, device, bias_code;
    int64_t n, k;
    bool allow_fused, allow_ws;
    std::shared_ptr<Unit>* unit;  // nullptr: no weight prefetch from this launch
    GroupObj* group;              // strong; nullptr: a layer on its own
    int idx;
    long epoch;
    std::vector<std::pair<int64_t, int>>* fused_yes;  // (m, dtype code) the one-launch route takes
    std::vector<std::pair<int64_t, int>>* fused_no;
};

PyObject *s_parameters, *s_buffers;

int64_t param_version(const at::Tensor& t) { return t.is_inference() ? -1 : (int64_t)t._version(); }  // linear._param_version

// linear._attr(mod, name) on the module's instance dictionary; borrowed reference (Py_None when absent)
PyObject* module_attr(PyObject* dict, PyObject* params, PyObject* name) {
    PyObject* t = params ? PyDict_GetItem(params, name) : nullptr;
    if (t) return t;
    t = PyDict_GetItem(dict, name);
    if (t) return t;
    PyObject* buffers = PyDict_GetItem(dict, s_buffers);
    t = buffers ? PyDict_GetItem(buffers, name) : nullptr;
    return t ? t : Py_None;
}

// linear._state's validity check (+ the bias, which the Python forward reads afresh on every call)
bool keys_current(PlanObj* p, PyObject* mod) {
    PyObject** dp = _PyObject_GetDictPtr(mod);
    if (!dp || !*dp) return false;
    PyObject* params = PyDict_GetItem(*dp, s_parameters);
    for (const KeyEnt& e : *p->keys) {
        PyObject* t = module_attr(*dp, params, e.name);
        if (t != e.ref) return false;
        if (t == Py_None) continue;
        const at::Tensor& tt = THPVariable_Unpack(t);
        if (tt.data_ptr() != e.ptr || param_version(tt) != e.version) return false;
    }
    return true;
}

PyObject* plan_call(PyObject* self, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
    PlanObj* p = (PlanObj*)self;
    if (PyVectorcall_NARGS(nargsf) != 2 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
        PyErr_SetString(PyExc_TypeError, "Plan(mod, input)");
        return nullptr;
    }
    PyObject* mod = args[0];
    PyObject* in = args[1];
    if (p->epoch != g_epoch.load(std::memory_order_relaxed)) Py_RETURN_FALSE;
    if (!THPVariable_CheckExact(in)) Py_RETURN_NONE;  // (tensor subclasses -- fake tensors, wrappers -- take the Python forward)
    try {
        const at::Tensor& x = THPVariable_Unpack(in);
        if (!x.is_cuda() || x.device().index() != p->device || x.dim() < 1 || x.size(-1) != p->k) Py_RETURN_NONE;
        const int code = float_code(x.scalar_type());
        if (code != SDNQ_BF16 && code != SDNQ_F16) Py_RETURN_NONE;
        const int64_t m = x.numel() / p->k;
        if (m < 32) Py_RETURN_NONE;  // linear_int8.py:102-103 (and M == 0): the Python forward's float branch
        if ((int)c10::hip::current_device() != p->device) Py_RETURN_NONE;
        if (!keys_current(p, mod)) Py_RETURN_FALSE;
        if (p->group) {  // ProjectionGroup.forward up to _claim; a miss (this member is the first one called) computes in Python
            if (t_no_reuse) Py_RETURN_NONE;
            PyObject* y = group_claim(p->group, p->idx, in, x);
            if (y) { g_plan_calls.fetch_add(1, std::memory_order_relaxed); return y; }
            if (PyErr_Occurred()) return nullptr;
            Py_RETURN_NONE;
        }
        at::Tensor x2 = x.dim() == 2 ? x : x.reshape({-1, p->k});
        if (x2.stride(-1) != 1 || (x2.stride(0) * (int64_t)x2.element_size()) % 16) x2 = x2.contiguous();
        void* stream = (void*)c10::hip::getCurrentHIPStream((c10::DeviceIndex)p->device).stream();
        bool fused = false;
        if (p->allow_fused && x2.stride(0) * 128 < ((int64_t)1 << 31)) {  // ops.linear_w8a8_fused_supported, memoized per (M, dtype)
            const std::pair<int64_t, int> q{m, code};
            bool known = false;
            for (auto& e : *p->fused_yes) if (e == q) { fused = true; known = true; break; }
            if (!known) for (auto& e : *p->fused_no) if (e == q) { known = true; break; }
            if (!known) {
                fused = f_fused_supported(p->mm, code, code, m, p->n, p->k) != 0;
                auto* v = fused ? p->fused_yes : p->fused_no;
                if (v->size() < 64) v->push_back(q);
            }
        }
        if (!fused) {
            if (!p->allow_ws) Py_RETURN_NONE;
            unsigned long long cap = 0;
            if (f_capture_id(stream, &cap) != SDNQ_OK || cap != 0) Py_RETURN_NONE;  // (a captured launch gets a scratch tensor of the graph's pool: ops.linear_w8a8_ws)
        }
        if (p->unit) unit_launch(*p->unit);
        at::Tensor out = at::empty({m, p->n}, x.options());
        const void* bias = p->bias->defined() ? p->bias->data_ptr() : nullptr;
        int rc;
        if (fused) {
            rc = f_fused(p->mm, x2.data_ptr(), code, m, p->k, x2.stride(0), p->wq->data_ptr(), (const float*)p->ws->data_ptr(), bias, p->bias_code,
                         out.data_ptr(), code, p->n, stream);
            g_fused_calls.fetch_add(1, std::memory_order_relaxed);
        } else {
            const int64_t xq_bytes = (m * p->k + 255) & ~(int64_t)255;
            const uintptr_t base = workspace(x, stream, xq_bytes + 4 * m + 512);
            rc = f_linear_w8a8(p->mm, x2.data_ptr(), code, m, p->k, x2.stride(0), p->had, (void*)base, (float*)(base + (uintptr_t)xq_bytes), p->wq->data_ptr(),
                               (const float*)p->ws->data_ptr(), bias, p->bias_code, out.data_ptr(), code, p->n, stream);
        }
        if (rc != SDNQ_OK) Py_RETURN_NONE;  // (nothing was launched: the Python forward repeats the call and reports the status)
        g_plan_calls.fetch_add(1, std::memory_order_relaxed);
        if (x.dim() == 2) return THPVariable_Wrap(out);
        std::vector<int64_t> shape = x.sizes().vec();
        shape.back() = p->n;
        return THPVariable_Wrap(out.view(shape));
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
}

PyObject* plan_new(PyTypeObject* type, PyObject* args, PyObject* kw) {
    static const char* kwlist[] = {"names", "refs", "mm", "n", "k", "wq", "ws", "bias", "had", "allow_fused", "allow_ws", "unit", "group", "idx", nullptr};
    PyObject *names, *refs, *wq = Py_None, *ws = Py_None, *bias = Py_None, *unit = Py_None, *group = Py_None;
    int mm = 0, had = 0, allow_fused = 0, allow_ws = 0, idx = 0;
    long long n = 0, k = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OOiLL|OOOippOOi", (char**)kwlist, &names, &refs, &mm, &n, &k, &wq, &ws, &bias, &had, &allow_fused, &allow_ws, &unit,
                                     &group, &idx))
        return nullptr;
    if (!PyTuple_Check(names) || !PyTuple_Check(refs) || PyTuple_GET_SIZE(names) != PyTuple_GET_SIZE(refs)) {
        PyErr_SetString(PyExc_TypeError, "names and refs must be tuples of one length");
        return nullptr;
    }
    if (unit != Py_None && !PyObject_TypeCheck(unit, &UnitType)) { PyErr_SetString(PyExc_TypeError, "unit must be a Unit"); return nullptr; }
    if (group != Py_None && !PyObject_TypeCheck(group, &GroupType)) { PyErr_SetString(PyExc_TypeError, "group must be a Group"); return nullptr; }
    if (group == Py_None && (!THPVariable_Check(wq) || !THPVariable_Check(ws))) { PyErr_SetString(PyExc_TypeError, "wq / ws must be tensors"); return nullptr; }
    if (bias != Py_None && !THPVariable_Check(bias)) { PyErr_SetString(PyExc_TypeError, "bias must be a tensor or None"); return nullptr; }
    PlanObj* p = (PlanObj*)type->tp_alloc(type, 0);
    if (!p) return nullptr;
    p->vc = plan_call;
    p->keys = new std::vector<KeyEnt>();
    p->wq = new at::Tensor();
    p->ws = new at::Tensor();
    p->bias = new at::Tensor();
    p->unit = nullptr;
    p->group = nullptr;
    p->fused_yes = new std::vector<std::pair<int64_t, int>>();
    p->fused_no = new std::vector<std::pair<int64_t, int>>();
    p->mm = mm; p->had = had; p->n = n; p->k = k; p->idx = idx; p->device = -1; p->bias_code = 0;
    p->allow_fused = allow_fused != 0;
    p->allow_ws = allow_ws != 0;
    p->epoch = g_epoch.load();
    try {
        for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(names); ++i) {
            PyObject* name = PyTuple_GET_ITEM(names, i);
            PyObject* ref = PyTuple_GET_ITEM(refs, i);
            if (!PyUnicode_Check(name) || (ref != Py_None && !THPVariable_Check(ref))) {
                PyErr_SetString(PyExc_TypeError, "names must be str, refs tensors or None");
                Py_DECREF(p);
                return nullptr;
            }
            KeyEnt e{name, ref, nullptr, 0};
            Py_INCREF(name);
            PyUnicode_InternInPlace(&e.name);
            Py_INCREF(ref);
            if (ref != Py_None) {
                const at::Tensor& t = THPVariable_Unpack(ref);
                e.ptr = t.data_ptr();
                e.version = param_version(t);
                if (p->device < 0 && t.is_cuda()) p->device = (int)t.device().index();
            }
            p->keys->push_back(e);
        }
        if (group != Py_None) {
            Py_INCREF(group);
            p->group = (GroupObj*)group;
        } else {
            *p->wq = THPVariable_Unpack(wq);
            *p->ws = THPVariable_Unpack(ws);
            if (!p->wq->is_cuda() || !p->ws->is_cuda() || p->ws->scalar_type() != at::kFloat || !p->ws->is_contiguous()) {
                PyErr_SetString(PyExc_ValueError, "wq / ws must be device tensors, ws contiguous float32");
                Py_DECREF(p);
                return nullptr;
            }
            p->device = (int)p->wq->device().index();
            if (bias != Py_None) {
                *p->bias = THPVariable_Unpack(bias);
                p->bias_code = float_code(p->bias->scalar_type());
                if (!p->bias->is_cuda() || !p->bias->is_contiguous() || p->bias_code < 0 || p->bias->numel() != n) {
                    PyErr_SetString(PyExc_ValueError, "bias must be a contiguous device tensor of N floats");
                    Py_DECREF(p);
                    return nullptr;
                }
            }
            if (unit != Py_None) p->unit = new std::shared_ptr<Unit>(*((UnitObj*)unit)->u);
        }
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        Py_DECREF(p);
        return nullptr;
    }
    if (p->device < 0) {
        PyErr_SetString(PyExc_ValueError, "a plan needs parameters on a device");
        Py_DECREF(p);
        return nullptr;
    }
    return (PyObject*)p;
}

void plan_dealloc(PlanObj* p) {
    if (p->keys) {
        for (KeyEnt& e : *p->keys) { Py_XDECREF(e.name); Py_XDECREF(e.ref); }
        delete p->keys;
    }
    delete p->wq;
    delete p->ws;
    delete p->bias;
    delete p->unit;
    Py_XDECREF((PyObject*)p->group);
    delete p->fused_yes;
    delete p->fused_no;
    Py_TYPE(p)->tp_free((PyObject*)p);
}
PyMemberDef plan_members[] = {{"mm", T_INT, offsetof(PlanObj, mm), READONLY, "matmul dtype code of the forward this plan restates"},
                              {nullptr, 0, 0, 0, nullptr}};
PyTypeObject PlanType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---------------------------------------------------------------------------------------------------------------------------
// module
// ---------------------------------------------------------------------------------------------------------------------------
PyObject* m_init(PyObject*, PyObject* arg) {
    const char* path = PyUnicode_AsUTF8(arg);
    if (!path) return nullptr;
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { PyErr_Format(PyExc_OSError, "dlopen(%s): %s", path, dlerror()); return nullptr; }
#define RESOLVE(var, name)                                                                                          \
    var = (decltype(var))dlsym(h, #name);                                                                           \
    if (!var) { PyErr_Format(PyExc_OSError, "%s does not export " #name, path); return nullptr; }
    RESOLVE(f_linear_w8a8, sdnq_hip_linear_w8a8)
    RESOLVE(f_fused, sdnq_hip_linear_w8a8_fused)
    RESOLVE(f_fused_supported, sdnq_hip_linear_w8a8_fused_supported)
    RESOLVE(f_hint, sdnq_hip_prefetch_hint)
    RESOLVE(f_capture_id, sdnq_hip_stream_capture_id)
#undef RESOLVE
    Py_RETURN_NONE;
}
PyObject* m_bump_epoch(PyObject*, PyObject*) { return PyLong_FromLong(g_epoch.fetch_add(1) + 1); }
PyObject* m_set_no_reuse(PyObject*, PyObject* arg) {
    long v = PyLong_AsLong(arg);
    if (v == -1 && PyErr_Occurred()) return nullptr;
    t_no_reuse = (int)v;
    Py_RETURN_NONE;
}
PyObject* m_chain_reset(PyObject*, PyObject*) {
    t_prev.reset();
    Py_RETURN_NONE;
}
PyObject* m_last_hint(PyObject*, PyObject*) {  // (count, p0, b0, p1, b1, p2, b2, p3, b3) of this thread
    PyObject* t = PyTuple_New(9);
    if (!t) return nullptr;
    for (int i = 0; i < 9; ++i) PyTuple_SET_ITEM(t, i, PyLong_FromLongLong(t_last_hint[i]));
    return t;
}
PyObject* m_fused_calls(PyObject*, PyObject*) { return PyLong_FromLong(g_fused_calls.load()); }
PyObject* m_plan_calls(PyObject*, PyObject*) { return PyLong_FromLong(g_plan_calls.load()); }
PyObject* m_reset_counters(PyObject*, PyObject*) {
    g_fused_calls = 0;
    g_plan_calls = 0;
    Py_RETURN_NONE;
}

PyMethodDef module_methods[] = {
    {"init", m_init, METH_O, "init(path of libsdnq_hip.so): resolve the entry points the plans launch"},
    {"bump_epoch", m_bump_epoch, METH_NOARGS, "every plan made so far is stale from now on"},
    {"set_no_reuse", m_set_no_reuse, METH_O, "this thread's identity_reuse_disabled() depth"},
    {"chain_reset", m_chain_reset, METH_NOARGS, "this thread's prefetch chain forgets its last unit"},
    {"last_hint", m_last_hint, METH_NOARGS, "(hints handed to the library by this thread, then the ranges of the last one)"},
    {"fused_calls", m_fused_calls, METH_NOARGS, "plan calls that took the one-launch route"},
    {"plan_calls", m_plan_calls, METH_NOARGS, "calls a plan carried"},
    {"reset_counters", m_reset_counters, METH_NOARGS, ""},
    {nullptr, nullptr, 0, nullptr}};
struct PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "_fastpath", "host-side fast path of the eager SDNQ Linear forward (see csrc/fastpath.cpp)", -1, module_methods};

}  // namespace

PyMODINIT_FUNC PyInit__fastpath(void) {
    s_parameters = PyUnicode_InternFromString("_parameters");
    s_buffers = PyUnicode_InternFromString("_buffers");

    UnitType.tp_name = "sdnq_amd._fastpath.Unit";
    UnitType.tp_basicsize = sizeof(UnitObj);
    UnitType.tp_flags = Py_TPFLAGS_DEFAULT;
    UnitType.tp_new = unit_new;
    UnitType.tp_dealloc = (destructor)unit_dealloc;
    UnitType.tp_methods = unit_methods;

    GroupType.tp_name = "sdnq_amd._fastpath.Group";
    GroupType.tp_basicsize = sizeof(GroupObj);
    GroupType.tp_flags = Py_TPFLAGS_DEFAULT;
    GroupType.tp_new = group_new;
    GroupType.tp_dealloc = (destructor)group_dealloc;
    GroupType.tp_methods = group_methods;
    GroupType.tp_members = group_members;

    PlanType.tp_name = "sdnq_amd._fastpath.Plan";
    PlanType.tp_basicsize = sizeof(PlanObj);
    PlanType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL;
    PlanType.tp_new = plan_new;
    PlanType.tp_dealloc = (destructor)plan_dealloc;
    PlanType.tp_members = plan_members;
    PlanType.tp_call = PyVectorcall_Call;
    PlanType.tp_vectorcall_offset = offsetof(PlanObj, vc);

    if (PyType_Ready(&UnitType) < 0 || PyType_Ready(&GroupType) < 0 || PyType_Ready(&PlanType) < 0) return nullptr;
    PyObject* m = PyModule_Create(&module_def);
    if (!m) return nullptr;
    Py_INCREF(&UnitType);
    Py_INCREF(&GroupType);
    Py_INCREF(&PlanType);
    PyModule_AddObject(m, "Unit", (PyObject*)&UnitType);
    PyModule_AddObject(m, "Group", (PyObject*)&GroupType);
    PyModule_AddObject(m, "Plan", (PyObject*)&PlanType);
    return m;
}
