/* _spmx_py -- the two host-side loops of the Python wrapper that do not belong in Python: turning list[str] into the
 * (pointer, length) views spmx_encode_batch_views takes (no packed copy: a str's UTF-8 form is read where CPython
 * keeps it), and turning the CSR that comes back into list[list[int]] -- what the reference's SWIG layer does in C++
 * (python/src/sentencepiece/sentencepiece.i:439-446, the std::vector<std::vector<int>> typemap).  The GIL is released
 * around the device call.  The library is reached through function pointers handed over by the ctypes binding
 * (sentencepiece_amd/_capi.py), so this module has no link-time dependency on libspmx.so. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { const char *data; uint64_t len; } view_t;
typedef int (*views_fn)(void *h, const view_t *views, uint64_t n, int32_t **ids, uint64_t **id_offsets, uint8_t **status,
                        uint64_t *n_failed);
typedef void (*free_fn)(void *p);

/* encode_views(fn_addr, handle_addr, seq) -> (ids_addr, offsets_addr, n, total).  seq: list / tuple of str or bytes.
 * The two arrays belong to the library (spmx_free). */
static PyObject *encode_views(PyObject *self, PyObject *args) {
  unsigned long long fn_addr = 0, h_addr = 0;
  PyObject *seq = NULL;
  if (!PyArg_ParseTuple(args, "KKO", &fn_addr, &h_addr, &seq)) return NULL;
  /* a TUPLE of our own: PySequence_Fast hands a list argument back as it is, and the views below point into str objects
   * that only that list keeps alive -- another thread could drop them while the GIL is released around the device call */
  PyObject *fast = PySequence_Tuple(seq);
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  view_t *views = (view_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(view_t));
  if (!views) { Py_DECREF(fast); return PyErr_NoMemory(); }
  PyObject **items = PySequence_Fast_ITEMS(fast);
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *o = items[i];
    Py_ssize_t len = 0;
    const char *p = NULL;
    if (PyUnicode_Check(o)) {
      p = PyUnicode_AsUTF8AndSize(o, &len);          /* cached in the object; an ASCII str is not even copied */
      if (!p) { free(views); Py_DECREF(fast); return NULL; }
    } else if (PyBytes_Check(o)) {
      char *q = NULL;
      if (PyBytes_AsStringAndSize(o, &q, &len) < 0) { free(views); Py_DECREF(fast); return NULL; }
      p = q;
    } else {
      free(views); Py_DECREF(fast);
      PyErr_SetString(PyExc_TypeError, "sentences must be str or bytes");
      return NULL;
    }
    views[i].data = p;
    views[i].len = (uint64_t)len;
  }
  int32_t *ids = NULL;
  uint64_t *offs = NULL;
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = ((views_fn)(uintptr_t)fn_addr)((void *)(uintptr_t)h_addr, views, (uint64_t)n, &ids, &offs, NULL, NULL);
  Py_END_ALLOW_THREADS
  free(views);
  Py_DECREF(fast);
  if (rc != 0) return Py_BuildValue("(KKni)", 0ULL, 0ULL, (Py_ssize_t)0, rc);
  return Py_BuildValue("(KKnK)", (unsigned long long)(uintptr_t)ids, (unsigned long long)(uintptr_t)offs, n,
                       (unsigned long long)(offs ? offs[n] : 0));
}

/* One int object per piece id, made on first use and shared by every list built afterwards: a list of ids then costs
 * a pointer store and a reference count per id instead of an object allocation (~20 ns each, which at ~28 ids per
 * sentence was 95 % of the list form's host time).  Ids are vocabulary indices, so the table is bounded by the largest
 * vocabulary seen (capped at 2^21 entries; ids beyond it get fresh objects). */
#define SPMX_PY_ID_CACHE_MAX (1 << 21)
static PyObject **g_id_cache = NULL;
static Py_ssize_t g_id_cache_n = 0;

static int grow_id_cache(Py_ssize_t want) {
  if (want > SPMX_PY_ID_CACHE_MAX) want = SPMX_PY_ID_CACHE_MAX;
  if (want <= g_id_cache_n) return 0;
  Py_ssize_t cap = g_id_cache_n ? g_id_cache_n : 1024;
  while (cap < want) cap *= 2;
  PyObject **t = (PyObject **)realloc(g_id_cache, (size_t)cap * sizeof(PyObject *));
  if (!t) { PyErr_NoMemory(); return -1; }
  memset(t + g_id_cache_n, 0, (size_t)(cap - g_id_cache_n) * sizeof(PyObject *));
  g_id_cache = t;
  g_id_cache_n = cap;
  return 0;
}

/* csr_to_lists(ids_addr, offsets_addr, n) -> list[list[int]] */
static PyObject *csr_to_lists(PyObject *self, PyObject *args) {
  unsigned long long ids_addr = 0, offs_addr = 0;
  Py_ssize_t n = 0;
  if (!PyArg_ParseTuple(args, "KKn", &ids_addr, &offs_addr, &n)) return NULL;
  const int32_t *ids = (const int32_t *)(uintptr_t)ids_addr;
  const uint64_t *offs = (const uint64_t *)(uintptr_t)offs_addr;
  PyObject *outer = PyList_New(n);
  if (!outer) return NULL;
  /* n new container objects would run the cycle collector every few hundred lists, each pass walking what has been
   * built so far (measured: 2.0 s instead of 0.35 s per million sentences); none of these lists can be in a cycle */
#if PY_VERSION_HEX >= 0x030A0000
  const int gc_was_on = PyGC_Disable();
#else
  const int gc_was_on = 0;                  /* (PyGC_Disable / PyGC_Enable exist from Python 3.10) */
#endif
  PyObject *ret = outer;
  for (Py_ssize_t i = 0; i < n && ret; ++i) {
    const uint64_t b = offs[i], e = offs[i + 1];
    PyObject *inner = PyList_New((Py_ssize_t)(e - b));
    if (!inner) { ret = NULL; break; }
    PyList_SET_ITEM(outer, i, inner);      /* (PyList_New zero-fills the items: a half-built list is safe to release) */
    for (uint64_t k = b; k < e; ++k) {
      const long id = (long)ids[k];
      PyObject *v;
      if (id >= 0 && id < SPMX_PY_ID_CACHE_MAX) {
        if (id >= g_id_cache_n && grow_id_cache(id + 1) < 0) { ret = NULL; break; }
        v = g_id_cache[id];
        if (!v) {
          v = PyLong_FromLong(id);
          if (!v) { ret = NULL; break; }
          g_id_cache[id] = v; /* the table keeps one reference for the life of the module */
        }
        Py_INCREF(v);
      } else {
        v = PyLong_FromLong(id);
        if (!v) { ret = NULL; break; }
      }
      PyList_SET_ITEM(inner, (Py_ssize_t)(k - b), v);
    }
  }
#if PY_VERSION_HEX >= 0x030A0000
  if (gc_was_on) PyGC_Enable();
#endif
  if (!ret) Py_DECREF(outer); /* the lists filled so far hold NULL or owned references only */
  return ret;
}

static PyMethodDef methods[] = {
    {"encode_views", encode_views, METH_VARARGS, "list[str|bytes] -> CSR held by the library (GIL released around the device call)"},
    {"csr_to_lists", csr_to_lists, METH_VARARGS, "CSR -> list[list[int]]"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_spmx_py", "host-side loops of the Python wrapper", -1, methods};
PyMODINIT_FUNC PyInit__spmx_py(void) { return PyModule_Create(&moddef); }
