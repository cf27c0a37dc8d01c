// oracle/rvo2_module/rvo2module.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// A CPython extension named `rvo2` exporting `PyRVOSimulator` with the surface the reference
// binds (gym_collision_avoidance/envs/policies/RVOPolicy.py:25-28 ctor kwargs, :46 addAgent,
// :70-74 setAgent{MaxSpeed,Radius,Position,Velocity,PrefVelocity}, :86-90 setAgentCollabCoeff,
// :93 doStep, :96 getAgentPosition), so the UNMODIFIED RVOPolicy.py runs against
// oracle/orca_ref.h.  The real module (mit-acl/Python-RVO2, Cython over the C++ RVO2 library) is
// an empty, un-fetchable submodule in /root/reference -- see the header of orca_ref.h
// ("parity unpinned").  All values cross the boundary as C `float`, like the Cython wrapper.
//
// doStep() recomputes EVERY agent's velocity (as the library does), although RVOPolicy only reads
// the ego agent's new position: this keeps the CPU-baseline cost structure honest.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <vector>

#include "../orca_ref.h"

namespace {

struct Sim {
  PyObject_HEAD
  float time_step, neighbor_dist, time_horizon, def_radius, def_max_speed;
  size_t max_neighbors;
  float global_time;
  std::vector<orca_ref::Body>* agents;
};

int get_xy(PyObject* o, float* x, float* y) {
  PyObject* seq = PySequence_Fast(o, "expected an (x, y) pair");
  if (!seq) return -1;
  if (PySequence_Fast_GET_SIZE(seq) != 2) {
    Py_DECREF(seq);
    PyErr_SetString(PyExc_ValueError, "expected an (x, y) pair");
    return -1;
  }
  const double dx = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(seq, 0));
  const double dy = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(seq, 1));
  Py_DECREF(seq);
  if (PyErr_Occurred()) return -1;
  *x = static_cast<float>(dx);
  *y = static_cast<float>(dy);
  return 0;
}

orca_ref::Body* agent_at(Sim* s, Py_ssize_t i) {
  if (i < 0 || static_cast<size_t>(i) >= s->agents->size()) {
    PyErr_SetString(PyExc_RuntimeError, "rvo2: no such agent");  // Cython wrapper raises on a bad index
    return nullptr;
  }
  return &(*s->agents)[static_cast<size_t>(i)];
}

int sim_init(PyObject* self, PyObject* args, PyObject* kw) {
  Sim* s = reinterpret_cast<Sim*>(self);
  static const char* names[] = {"timeStep", "neighborDist", "maxNeighbors", "timeHorizon", "timeHorizonObst",
                                "radius",   "maxSpeed",     "velocity",     nullptr};
  float ts, nd, th, tho, r, ms;
  Py_ssize_t mn;
  PyObject* vel = nullptr;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "ffnffff|O", const_cast<char**>(names), &ts, &nd, &mn, &th, &tho, &r,
                                   &ms, &vel))
    return -1;
  s->time_step = ts;
  s->neighbor_dist = nd;
  s->max_neighbors = static_cast<size_t>(mn);
  s->time_horizon = th;
  s->def_radius = r;
  s->def_max_speed = ms;
  s->global_time = 0.0f;
  (void)tho;
  return 0;
}

PyObject* sim_new(PyTypeObject* t, PyObject*, PyObject*) {
  Sim* s = reinterpret_cast<Sim*>(t->tp_alloc(t, 0));
  if (s) s->agents = new std::vector<orca_ref::Body>();
  return reinterpret_cast<PyObject*>(s);
}

void sim_dealloc(PyObject* self) {
  Sim* s = reinterpret_cast<Sim*>(self);
  delete s->agents;
  Py_TYPE(self)->tp_free(self);
}

PyObject* add_agent(PyObject* self, PyObject* args, PyObject* kw) {
  Sim* s = reinterpret_cast<Sim*>(self);
  PyObject* pos;
  static const char* names[] = {"pos", nullptr};
  if (!PyArg_ParseTupleAndKeywords(args, kw, "O", const_cast<char**>(names), &pos)) return nullptr;
  orca_ref::Body b;
  if (get_xy(pos, &b.pos.x, &b.pos.y)) return nullptr;
  b.vel = orca_ref::mk(0.f, 0.f);
  b.pref = orca_ref::mk(0.f, 0.f);
  b.radius = s->def_radius;
  b.max_speed = s->def_max_speed;
  b.collab = 0.5f;
  s->agents->push_back(b);
  return PyLong_FromSize_t(s->agents->size() - 1);
}

#define SETTER_XY(NAME, FIELD)                                   \
  PyObject* NAME(PyObject* self, PyObject* args) {               \
    Py_ssize_t i;                                                \
    PyObject* v;                                                 \
    if (!PyArg_ParseTuple(args, "nO", &i, &v)) return nullptr;   \
    orca_ref::Body* b = agent_at(reinterpret_cast<Sim*>(self), i); \
    if (!b) return nullptr;                                      \
    if (get_xy(v, &b->FIELD.x, &b->FIELD.y)) return nullptr;     \
    Py_RETURN_NONE;                                              \
  }
#define SETTER_F(NAME, FIELD)                                    \
  PyObject* NAME(PyObject* self, PyObject* args) {               \
    Py_ssize_t i;                                                \
    float v;                                                     \
    if (!PyArg_ParseTuple(args, "nf", &i, &v)) return nullptr;   \
    orca_ref::Body* b = agent_at(reinterpret_cast<Sim*>(self), i); \
    if (!b) return nullptr;                                      \
    b->FIELD = v;                                                \
    Py_RETURN_NONE;                                              \
  }
#define GETTER_XY(NAME, FIELD)                                   \
  PyObject* NAME(PyObject* self, PyObject* args) {               \
    Py_ssize_t i;                                                \
    if (!PyArg_ParseTuple(args, "n", &i)) return nullptr;        \
    orca_ref::Body* b = agent_at(reinterpret_cast<Sim*>(self), i); \
    if (!b) return nullptr;                                      \
    return Py_BuildValue("(dd)", static_cast<double>(b->FIELD.x), static_cast<double>(b->FIELD.y)); \
  }

SETTER_XY(set_pos, pos)
SETTER_XY(set_vel, vel)
SETTER_XY(set_pref, pref)
SETTER_F(set_radius, radius)
SETTER_F(set_max_speed, max_speed)
SETTER_F(set_collab, collab)
GETTER_XY(get_pos, pos)
GETTER_XY(get_vel, vel)
GETTER_XY(get_pref, pref)

PyObject* do_step(PyObject* self, PyObject*) {
  Sim* s = reinterpret_cast<Sim*>(self);
  std::vector<orca_ref::Body>& a = *s->agents;
  const size_t n = a.size();
  std::vector<orca_ref::Vec> nv(n);
  for (size_t i = 0; i < n; ++i)
    nv[i] = orca_ref::new_velocity(a.data(), n, i, s->neighbor_dist, s->max_neighbors, s->time_horizon, s->time_step);
  for (size_t i = 0; i < n; ++i) {
    a[i].vel = nv[i];
    a[i].pos = orca_ref::advance(a[i].pos, nv[i], s->time_step);
  }
  s->global_time += s->time_step;
  Py_RETURN_NONE;
}

PyObject* num_agents(PyObject* self, PyObject*) { return PyLong_FromSize_t(reinterpret_cast<Sim*>(self)->agents->size()); }
PyObject* global_time(PyObject* self, PyObject*) { return PyFloat_FromDouble(reinterpret_cast<Sim*>(self)->global_time); }

PyMethodDef sim_methods[] = {
    {"addAgent", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(add_agent)), METH_VARARGS | METH_KEYWORDS, ""},
    {"setAgentPosition", set_pos, METH_VARARGS, ""},
    {"setAgentVelocity", set_vel, METH_VARARGS, ""},
    {"setAgentPrefVelocity", set_pref, METH_VARARGS, ""},
    {"setAgentRadius", set_radius, METH_VARARGS, ""},
    {"setAgentMaxSpeed", set_max_speed, METH_VARARGS, ""},
    {"setAgentCollabCoeff", set_collab, METH_VARARGS, ""},
    {"getAgentPosition", get_pos, METH_VARARGS, ""},
    {"getAgentVelocity", get_vel, METH_VARARGS, ""},
    {"getAgentPrefVelocity", get_pref, METH_VARARGS, ""},
    {"doStep", do_step, METH_NOARGS, ""},
    {"getNumAgents", num_agents, METH_NOARGS, ""},
    {"getGlobalTime", global_time, METH_NOARGS, ""},
    {nullptr, nullptr, 0, nullptr}};

PyTypeObject SimType = {PyVarObject_HEAD_INIT(nullptr, 0)};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "rvo2", "CPU-oracle restatement of the rvo2 binding (see orca_ref.h)", -1,
                      nullptr,               nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit_rvo2(void) {
  SimType.tp_name = "rvo2.PyRVOSimulator";
  SimType.tp_basicsize = sizeof(Sim);
  SimType.tp_flags = Py_TPFLAGS_DEFAULT;
  SimType.tp_new = sim_new;
  SimType.tp_init = sim_init;
  SimType.tp_dealloc = sim_dealloc;
  SimType.tp_methods = sim_methods;
  if (PyType_Ready(&SimType) < 0) return nullptr;
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return nullptr;
  Py_INCREF(&SimType);
  PyModule_AddObject(m, "PyRVOSimulator", reinterpret_cast<PyObject*>(&SimType));
  PyModule_AddStringConstant(m, "__oracle__", "restated from the published RVO2 algorithm; parity unpinned");
  return m;
}
