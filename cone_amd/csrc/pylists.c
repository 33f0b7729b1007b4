/* Host-side tail of a step: the kept rows of stage C, (T, nq, A, 5) doubles + counts, become the reference's
 * `predicted_times` lists (cone/inference.py:141-166: per query a list of [st, ed, proposal, matching, fused] rows) in ONE
 * pass over the buffer -- instead of tensor.tolist() (which builds a (T, nq, A, 5) nest that is then re-sliced) and a
 * Python loop of 3 x nq dict assignments.  Plain CPython C API, loaded with ctypes.PyDLL (the calls hold the GIL); not
 * part of the device library.  Build: cone_amd/build.py (gcc, no link against libpython: the interpreter exports the
 * symbols). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* items: list of nq dicts; rows: nq * A * 5 doubles; n: nq counts (<= A).  Sets item[key] = [[5 floats] * n[q]]. */
PyObject* cone_fill_predicted_times(PyObject* items, const double* rows, const int32_t* n, Py_ssize_t nq, Py_ssize_t A,
                                    PyObject* key) {
    if (!PyList_Check(items) || PyList_GET_SIZE(items) != nq) {
        PyErr_SetString(PyExc_ValueError, "cone_fill_predicted_times: items must be a list of nq dicts");
        return NULL;
    }
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject* item = PyList_GET_ITEM(items, q);
        Py_ssize_t k = n[q] < 0 ? 0 : (n[q] > A ? A : n[q]);
        PyObject* pt = PyList_New(k);
        if (!pt) return NULL;
        const double* r = rows + (size_t)q * A * 5;
        for (Py_ssize_t i = 0; i < k; ++i) {
            PyObject* row = PyList_New(5);
            if (!row) { Py_DECREF(pt); return NULL; }
            for (int c = 0; c < 5; ++c) {
                PyObject* f = PyFloat_FromDouble(r[i * 5 + c]);
                if (!f) { Py_DECREF(row); Py_DECREF(pt); return NULL; }
                PyList_SET_ITEM(row, c, f);
            }
            PyList_SET_ITEM(pt, i, row);
        }
        if (PyDict_SetItem(item, key, pt) < 0) { Py_DECREF(pt); return NULL; }
        Py_DECREF(pt);
    }
    Py_RETURN_NONE;
}
