/* _pytext: where the code units of a list of str lie, for tt_tok_encode_units (include/tt.h).  The only CPython-API code of
 * the package, and no part of libtt.so (whose C ABI has no Python in it): twotowermlretrieval_amd/build.py compiles it with the
 * interpreter's own headers; tokenizer.encode_batch works without it (one join + encode per batch instead).
 * What it replaces: "\0".join(texts).encode("ascii") under the interpreter lock -- 2.7 ms per 16 k passages, two passes over the
 * text and a fresh 7 MB allocation -- by one pass over the list's object pointers (~0.1 ms). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* gather(texts: list | tuple, ptrs_addr: int, lens_addr: int, units_addr: int) -> (n_ok, total_units, n_beyond_ascii)
 * ptrs_addr / lens_addr / units_addr: addresses of caller-owned arrays of len(texts) pointers / int64 / uint8.  For every str:
 * where its code units lie, how many there are (= code points: CPython stores one per unit), and the unit size for
 * tt_tok_encode_units: 0 = ASCII bytes, 1 / 2 / 4 = Latin-1 / UCS-2 / UCS-4 units.  Stops at the first item that is not a str
 * (n_ok < len(texts): the caller takes its general path for the batch).  The arrays point INTO the str objects: they are valid as
 * long as the caller keeps `texts` alive and unchanged. */
static PyObject *gather(PyObject *self, PyObject *args)
{
    PyObject *seq;
    unsigned long long pa, la, ua;
    (void)self;
    if (!PyArg_ParseTuple(args, "OKKK", &seq, &pa, &la, &ua))
        return NULL;
    if (!PyList_CheckExact(seq) && !PyTuple_CheckExact(seq)) {
        PyErr_SetString(PyExc_TypeError, "gather: a list or tuple of str");
        return NULL;
    }
    const void **ptrs = (const void **)(uintptr_t)pa;
    int64_t *lens = (int64_t *)(uintptr_t)la;
    uint8_t *units = (uint8_t *)(uintptr_t)ua;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    Py_ssize_t i = 0, beyond = 0;
    long long total = 0;
    for (; i < n; ++i) {
        PyObject *o = items[i];
        if (!PyUnicode_CheckExact(o) || PyUnicode_READY(o) != 0)
            break;
        ptrs[i] = (const void *)PyUnicode_DATA(o);
        lens[i] = (int64_t)PyUnicode_GET_LENGTH(o);
        if (PyUnicode_IS_ASCII(o)) {
            units[i] = 0;
        } else {
            units[i] = (uint8_t)PyUnicode_KIND(o); /* PyUnicode_1BYTE_KIND = 1, _2BYTE_ = 2, _4BYTE_ = 4 */
            ++beyond;
        }
        total += lens[i];
    }
    if (PyErr_Occurred())
        return NULL;
    return Py_BuildValue("nLn", i, total, beyond);
}

static PyMethodDef methods[] = {{"gather", gather, METH_VARARGS, "pointers, lengths and unit sizes of a list of str"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pytext", NULL, -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__pytext(void) { return PyModule_Create(&module); }
