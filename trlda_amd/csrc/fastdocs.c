/*
 * fastdocs.c -- CPython-3 extension `trlda_amd._fastdocs`: the list-of-lists-of-(id, count)
 * tuples that the reference's Python surface takes, flattened to CSR int32 in one pass in C.
 *
 * Counterpart of PyList_ToDocuments (reference python/src/ldainterface.cpp:152-190), which
 * walks the same structure with PyList_GetItem / PyArg_ParseTuple(word, "ii", ...) into
 * vector<vector<pair<int,int>>>.  This walks it once into three byte strings (indptr[B+1],
 * ids[nnz], cnts[nnz], native int32) that NumPy wraps without copying.
 *
 *   flatten(docs) -> (indptr, ids, cnts)   the fast path applied
 *                 -> None                  something is not a list / 2-tuple of ints in int32
 *                                          range: the caller's Python path then reports it
 *                                          with the reference's exceptions
 * Host-side format conversion only; nothing here computes.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *flatten(PyObject *self, PyObject *docs)
{
    (void)self;
    if (!PyList_CheckExact(docs))
        Py_RETURN_NONE;
    const Py_ssize_t B = PyList_GET_SIZE(docs);
    int64_t nnz = 0;
    for (Py_ssize_t d = 0; d < B; ++d) {
        PyObject *doc = PyList_GET_ITEM(docs, d);
        if (!PyList_CheckExact(doc))
            Py_RETURN_NONE;
        nnz += PyList_GET_SIZE(doc);
    }
    if (nnz >= INT32_MAX || B >= INT32_MAX)
        Py_RETURN_NONE;
    PyObject *indptr_b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)(B + 1) * 4);
    PyObject *ids_b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)nnz * 4);
    PyObject *cnts_b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)nnz * 4);
    if (!indptr_b || !ids_b || !cnts_b)
        goto fail_mem;
    {
        int32_t *indptr = (int32_t *)PyBytes_AS_STRING(indptr_b);
        int32_t *ids = (int32_t *)PyBytes_AS_STRING(ids_b);
        int32_t *cnts = (int32_t *)PyBytes_AS_STRING(cnts_b);
        int64_t pos = 0;
        indptr[0] = 0;
        for (Py_ssize_t d = 0; d < B; ++d) {
            PyObject *doc = PyList_GET_ITEM(docs, d);
            /* the lists are not ours: their length may have changed since the first pass */
            const Py_ssize_t n = PyList_GET_SIZE(doc);
            if (pos + n > nnz)
                goto not_fast;
            for (Py_ssize_t j = 0; j < n; ++j) {
                PyObject *word = PyList_GET_ITEM(doc, j);
                if (!PyTuple_CheckExact(word) || PyTuple_GET_SIZE(word) != 2)
                    goto not_fast;
                PyObject *a = PyTuple_GET_ITEM(word, 0), *b = PyTuple_GET_ITEM(word, 1);
                if (!PyLong_Check(a) || !PyLong_Check(b))
                    goto not_fast;
                int oa = 0, ob = 0;
                const long va = PyLong_AsLongAndOverflow(a, &oa);
                const long vb = PyLong_AsLongAndOverflow(b, &ob);
                if (oa || ob || va < INT32_MIN || va > INT32_MAX || vb < INT32_MIN || vb > INT32_MAX)
                    goto not_fast;
                ids[pos] = (int32_t)va;
                cnts[pos] = (int32_t)vb;
                ++pos;
            }
            indptr[d + 1] = (int32_t)pos;
        }
        if (pos != nnz)
            goto not_fast;
    }
    {
        PyObject *out = PyTuple_Pack(3, indptr_b, ids_b, cnts_b);
        Py_DECREF(indptr_b);
        Py_DECREF(ids_b);
        Py_DECREF(cnts_b);
        return out;
    }
not_fast:
    Py_DECREF(indptr_b);
    Py_DECREF(ids_b);
    Py_DECREF(cnts_b);
    Py_RETURN_NONE;
fail_mem:
    Py_XDECREF(indptr_b);
    Py_XDECREF(ids_b);
    Py_XDECREF(cnts_b);
    return PyErr_NoMemory();
}

/* tuples(indptr, ids, cnts) -> list of lists of (id, count) tuples: the inverse, for the text
 * loader's drop-in return type (reference python/utils/load_documents.py:40-45). */
static PyObject *tuples(PyObject *self, PyObject *args)
{
    (void)self;
    Py_buffer ip, ii, cc;
    if (!PyArg_ParseTuple(args, "y*y*y*", &ip, &ii, &cc))
        return NULL;
    PyObject *out = NULL;
    const Py_ssize_t B = ip.len / 4 - 1;
    const int32_t *indptr = (const int32_t *)ip.buf;
    const int32_t *ids = (const int32_t *)ii.buf, *cnts = (const int32_t *)cc.buf;
    if (ip.len < 4 || ii.len != cc.len || indptr[0] != 0 || (Py_ssize_t)indptr[B] * 4 != ii.len) {
        PyErr_SetString(PyExc_ValueError, "inconsistent CSR arrays");
        goto done;
    }
    out = PyList_New(B);
    if (!out)
        goto done;
    for (Py_ssize_t d = 0; d < B; ++d) {
        const int32_t p0 = indptr[d], p1 = indptr[d + 1];
        PyObject *doc = p1 >= p0 ? PyList_New(p1 - p0) : NULL;
        if (!doc) {
            if (p1 < p0)
                PyErr_SetString(PyExc_ValueError, "indptr must be non-decreasing");
            Py_CLEAR(out);
            goto done;
        }
        PyList_SET_ITEM(out, d, doc);
        for (int32_t p = p0; p < p1; ++p) {
            PyObject *a = PyLong_FromLong(ids[p]), *b = PyLong_FromLong(cnts[p]);
            PyObject *t = (a && b) ? PyTuple_Pack(2, a, b) : NULL;
            Py_XDECREF(a);
            Py_XDECREF(b);
            if (!t) {
                Py_CLEAR(out);
                goto done;
            }
            PyList_SET_ITEM(doc, p - p0, t);
        }
    }
done:
    PyBuffer_Release(&ip);
    PyBuffer_Release(&ii);
    PyBuffer_Release(&cc);
    return out;
}

static PyMethodDef methods[] = {
    {"flatten", flatten, METH_O,
     "flatten(docs) -> (indptr, ids, cnts) as int32 byte strings, or None if docs is not a list "
     "of lists of 2-tuples of ints in int32 range"},
    {"tuples", tuples, METH_VARARGS,
     "tuples(indptr, ids, cnts) -> list of lists of (id, count) tuples"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fastdocs",
                                    "CSR <-> list-of-tuples conversion for trlda_amd.documents", -1,
                                    methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastdocs(void) { return PyModule_Create(&module); }
