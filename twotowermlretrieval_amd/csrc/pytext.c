/* _pytext: the pointers and byte lengths of a list of str, for tt_tok_encode_ptrs (include/tt.h).  The only CPython-API code of
 * the package, and no part of libtt.so (whose C ABI has no Python in it): twotowermlretrieval_amd/build.py compiles it with the
 * interpreter's own headers; tokenizer.encode_batch works without it (one join + encode per batch instead).
 * What it replaces: "\0".join(texts).encode("ascii") under the interpreter lock -- 2.7 ms per 16 k passages, two passes over the
 * text and a fresh 7 MB allocation -- by one pass over the list's object pointers (~0.1 ms). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* gather(texts: list | tuple, ptrs_addr: int, lens_addr: int) -> (n_ok, total_bytes)
 * ptrs_addr / lens_addr: addresses of caller-owned arrays of len(texts) pointers / int64.  Stops at the first item that is not a
 * str in the compact ASCII form (n_ok < len(texts): the caller takes its general path for the batch).  The arrays point INTO the
 * str objects: they are valid as long as the caller keeps `texts` alive and unchanged. */
static PyObject *gather(PyObject *self, PyObject *args)
{
    PyObject *seq;
    unsigned long long pa, la;
    (void)self;
    if (!PyArg_ParseTuple(args, "OKK", &seq, &pa, &la))
        return NULL;
    if (!PyList_CheckExact(seq) && !PyTuple_CheckExact(seq)) {
        PyErr_SetString(PyExc_TypeError, "gather: a list or tuple of str");
        return NULL;
    }
    const char **ptrs = (const char **)(uintptr_t)pa;
    int64_t *lens = (int64_t *)(uintptr_t)la;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    Py_ssize_t i = 0;
    long long total = 0;
    for (; i < n; ++i) {
        PyObject *o = items[i];
        if (!PyUnicode_CheckExact(o) || PyUnicode_READY(o) != 0 || !PyUnicode_IS_COMPACT_ASCII(o))
            break;
        ptrs[i] = (const char *)PyUnicode_DATA(o);
        lens[i] = (int64_t)PyUnicode_GET_LENGTH(o);
        total += lens[i];
    }
    if (PyErr_Occurred())
        return NULL;
    return Py_BuildValue("nL", i, total);
}

static PyMethodDef methods[] = {{"gather", gather, METH_VARARGS, "pointers and lengths of a list of ASCII str"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pytext", NULL, -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__pytext(void) { return PyModule_Create(&module); }
