/* _mmvae_pylookup.so: one CPython-API helper of the step engine's host side (loaded with ctypes.PyDLL: the interpreter
 * lock is held).  The conditional layers map every cell's metadata value (a str) to the index of its condition block:
 * 512 dictionary look-ups per conditional layer and step, which cost 60 us per layer from the interpreter (np.fromiter
 * over a generator) in a host-bound program; here ~10 us.  Reference: ConditionalLayer.forward builds Python masks per
 * condition from the same column (components.py:365-413). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* out[i] = table[values[i]] for the n items of the list `values` (int values that fit int32).
 * Returns n when every key was found; otherwise the index of the first value that is not a key (>= 0, < n; nothing is
 * raised: the caller extends its table and calls again), or -1 with a Python exception set (wrong types / an
 * unhashable value / a table value that is not an int32). */
Py_ssize_t mmvae_py_lookup_i32(PyObject* table, PyObject* values, int32_t* out, Py_ssize_t n) {
    if (!PyDict_Check(table) || !PyList_Check(values) || PyList_GET_SIZE(values) != n || !out) {
        PyErr_SetString(PyExc_TypeError, "mmvae_py_lookup_i32(dict, list of n values, int32 buffer, n)");
        return -1;
    }
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* hit = PyDict_GetItemWithError(table, PyList_GET_ITEM(values, i)); /* borrowed */
        if (!hit) {
            if (PyErr_Occurred()) return -1;
            return i;
        }
        const long v = PyLong_AsLong(hit);
        if ((v == -1 && PyErr_Occurred()) || v < INT32_MIN || v > INT32_MAX) {
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_OverflowError, "table value does not fit int32");
            return -1;
        }
        out[i] = (int32_t)v;
    }
    return n;
}
