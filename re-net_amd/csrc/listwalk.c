/* Flattening of the reference's nested history lists (train.py:136-137 hands RENet.forward, per batch,
 *   hist   = [ [ndarray[k,2] (r, o) per step] per sequence ]      hist_t = [ [timestamp per step] per sequence ] )
 * into the four flat int64 arrays of graph.FlatHistory -- sequence lengths, neighbours per step, the neighbour column (o) and the
 * step timestamps -- in ONE pass over the Python objects.  The numpy formulation (graph.FlatHistory.from_lists: map(len), one
 * np.concatenate over ~8.6 k tiny arrays, np.fromiter over the timestamps) costs 2.4 ms per call on the MI355X box's host, twice per
 * step: half of the 10.4 ms an unmodified train.py step takes (profiles/r05_k_*).  Host-side utility only: no device work, and
 * graph.FlatHistory.from_lists falls back to the numpy formulation when this module is absent or declines an input.
 *
 * CPython C API + buffer protocol (no numpy headers): arrays must expose a 2-D buffer [k, 2] of 8-byte signed integers (any strides).
 * Returns four bytearrays (int64 little-endian host order) the caller wraps with np.frombuffer. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#ifdef RENET_LISTWALK_NUMPY            /* built with numpy's headers: ndarrays are read through their struct (a few ns per array) */
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t* p; Py_ssize_t n, cap; } vec;

static int vpush(vec* v, int64_t x) {
    if (v->n == v->cap) {
        Py_ssize_t nc = v->cap ? v->cap * 2 : 1024;
        int64_t* q = (int64_t*)realloc(v->p, (size_t)nc * sizeof(int64_t));
        if (!q) return -1;
        v->p = q; v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

static PyObject* to_bytearray(vec* v) {
    return PyByteArray_FromStringAndSize((const char*)v->p, v->n * (Py_ssize_t)sizeof(int64_t));
}

static PyObject* flatten(PyObject* self, PyObject* args) {
    PyObject *hist, *hist_t;
    if (!PyArg_ParseTuple(args, "OO", &hist, &hist_t)) return NULL;
    if (!PyList_Check(hist) || !PyList_Check(hist_t) || PyList_GET_SIZE(hist) != PyList_GET_SIZE(hist_t)) {
        PyErr_SetString(PyExc_TypeError, "flatten(hist, hist_t): two lists of equal length");
        return NULL;
    }
    vec lens = {0, 0, 0}, cnt = {0, 0, 0}, nbr = {0, 0, 0}, st = {0, 0, 0};
    const Py_ssize_t n = PyList_GET_SIZE(hist);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* h = PyList_GET_ITEM(hist, i);
        PyObject* ht = PyList_GET_ITEM(hist_t, i);
        if (!PyList_Check(h) || !PyList_Check(ht) || PyList_GET_SIZE(h) != PyList_GET_SIZE(ht)) {
            PyErr_SetString(PyExc_TypeError, "history entries must be lists of equal length");
            goto fail;
        }
        const Py_ssize_t L = PyList_GET_SIZE(h);
        if (vpush(&lens, (int64_t)L) < 0) goto nomem;
        for (Py_ssize_t j = 0; j < L; ++j) {
            PyObject* arr = PyList_GET_ITEM(h, j);
#ifdef RENET_LISTWALK_NUMPY
            if (PyArray_Check(arr) && PyArray_NDIM((PyArrayObject*)arr) == 2 && PyArray_DIM((PyArrayObject*)arr, 1) == 2 &&
                PyArray_TYPE((PyArrayObject*)arr) == NPY_INT64 && PyArray_ISNOTSWAPPED((PyArrayObject*)arr)) {
                PyArrayObject* a = (PyArrayObject*)arr;
                const Py_ssize_t k = PyArray_DIM(a, 0), s0 = PyArray_STRIDE(a, 0);
                const char* base = PyArray_BYTES(a) + PyArray_STRIDE(a, 1);
                int bad = vpush(&cnt, (int64_t)k) < 0;
                for (Py_ssize_t r = 0; r < k && !bad; ++r) {
                    int64_t x;
                    memcpy(&x, base + r * s0, sizeof(x));
                    bad = vpush(&nbr, x) < 0;
                }
                if (bad) goto nomem;
                const long long t = PyLong_AsLongLong(PyList_GET_ITEM(ht, j));
                if (t == -1 && PyErr_Occurred()) goto fail;
                if (vpush(&st, (int64_t)t) < 0) goto nomem;
                continue;
            }
#endif
            Py_buffer view;
            if (PyObject_GetBuffer(arr, &view, PyBUF_STRIDES | PyBUF_FORMAT) < 0) goto fail;
            const char* f = view.format ? view.format : "";
            while (*f == '<' || *f == '=' || *f == '@') ++f;
            const int ok = view.ndim == 2 && view.shape[1] == 2 && view.itemsize == 8 && (*f == 'l' || *f == 'q') && f[1] == 0;
            if (!ok) {
                PyBuffer_Release(&view);
                PyErr_SetString(PyExc_TypeError, "a history step must be an int64 array of shape [k, 2]");
                goto fail;
            }
            const Py_ssize_t k = view.shape[0];
            const char* base = (const char*)view.buf + view.strides[1];          /* column 1: the neighbour entity */
            int bad = vpush(&cnt, (int64_t)k) < 0;
            for (Py_ssize_t r = 0; r < k && !bad; ++r) {
                int64_t x;
                memcpy(&x, base + r * view.strides[0], sizeof(x));
                bad = vpush(&nbr, x) < 0;
            }
            PyBuffer_Release(&view);
            if (bad) goto nomem;
            const long long t = PyLong_AsLongLong(PyList_GET_ITEM(ht, j));        /* (numpy / torch scalars: via __index__) */
            if (t == -1 && PyErr_Occurred()) goto fail;
            if (vpush(&st, (int64_t)t) < 0) goto nomem;
        }
    }
    {
        PyObject *a = to_bytearray(&lens), *b = to_bytearray(&cnt), *c = to_bytearray(&nbr), *d = to_bytearray(&st);
        free(lens.p); free(cnt.p); free(nbr.p); free(st.p);
        if (!a || !b || !c || !d) { Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(d); return NULL; }
        PyObject* out = PyTuple_Pack(4, a, b, c, d);
        Py_DECREF(a); Py_DECREF(b); Py_DECREF(c); Py_DECREF(d);
        return out;
    }
nomem:
    PyErr_NoMemory();
fail:
    free(lens.p); free(cnt.p); free(nbr.p); free(st.p);
    return NULL;
}

static PyMethodDef methods[] = {
    {"flatten", flatten, METH_VARARGS, "flatten(hist, hist_t) -> (lens, neighbours per step, neighbour column, step timestamps) as int64 bytearrays"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_renet_listwalk", "nested history lists -> flat int64 arrays", -1, methods};

PyMODINIT_FUNC PyInit__renet_listwalk(void) {
#ifdef RENET_LISTWALK_NUMPY
    import_array();
#endif
    return PyModule_Create(&moddef);
}
