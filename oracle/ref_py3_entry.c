/*
 * ref_py3_entry.c -- Python-3 entry point for the UNMODIFIED reference module.
 *
 * TEST INFRASTRUCTURE ONLY (see pb_oracle.c).  Nothing from the reference is
 * copied into this repository: the translation unit below #includes
 * moira/bernoullimodule.c from where it lies under /root/reference (path given
 * by the Makefile as MOIRA_REF_SRC) and the output goes to oracle/_ref/ only.
 *
 * The reference targets the CPython-2 C API.  Python 3.10's Python.h (present
 * in this image) differs for this file in exactly two names:
 *   PyInt_AsLong   -- renamed PyLong_AsLong in Python 3 (same semantics);
 *   Py_InitModule  -- replaced by PyModule_Create + PyInit_<name>.
 * The first is aliased; the second is only used by the reference's
 * `initbernoulli` (bernoullimodule.c:122-125), which Python 3 never calls, so
 * it is compiled to a no-op and the module is registered below from the
 * reference's own method table (`module_methods`, bernoullimodule.c:117-120).
 * All arithmetic (prob_j_errors, sum_of_binomials, interpolate, test) is the
 * reference's, untouched.
 */
#include <Python.h>

#define PyInt_AsLong PyLong_AsLong
#define Py_InitModule(name, methods) ((void)0)
#undef PyMODINIT_FUNC
#define PyMODINIT_FUNC void

#include MOIRA_REF_SRC

static struct PyModuleDef moira_ref_def = {
    PyModuleDef_HEAD_INIT, "bernoulli", module_docstring, -1, module_methods,
    NULL, NULL, NULL, NULL
};

PyObject *PyInit_bernoulli(void) { return PyModule_Create(&moira_ref_def); }
