"""ctypes mirror of include/cerberus_mfp.h (POD descriptors and enums only)."""
import ctypes as C

MAX_DIM = 3
MAX_COMP = 10
MAX_ACTION_STATES = 8
MAX_TRACERS = 2

# cmfp_status
OK, ERR_INVALID, ERR_HIP, ERR_PHYSICS, ERR_NODEVICE = 0, -1, -2, -3, -4

STATE_HYDRO, STATE_FIELD = 0, 1

RECON = {"constant": 0, "minmod": 1, "vanLeer": 2, "MC": 3, "centre": 4, "O6": 5, "WENO": 6}
RECON_STENCIL = {0: 1, 1: 3, 2: 3, 3: 3, 4: 3, 5: 7, 6: 7}
RECON_NGROW = {0: 1, 1: 2, 2: 2, 3: 2, 4: 2, 5: 4, 6: 4}

FLUX = {"HLLC": 0, "HLLE": 1, "HLLE/HLLC": 2, "RankineHugoniot": 10}

BC_BOGUS, BC_REFLECT_ODD, BC_INT_DIR, BC_REFLECT_EVEN, BC_FOEXTRAP, BC_EXT_DIR, BC_HOEXTRAP = (
    -666, -1, 0, 1, 2, 3, 4)

ACTION = {"CTU": 0, "plasma5": 1, "Lorentz": 2, "collisions_rambo": 3}
SOLVER = {"explicit": 0, "implicit": 1}
VISCOSITY = {"Sutherland": 1, "PowerLaw": 2, "UserDefined": 3}
TIME_INTEGRATOR = {"euler": 0, "RK2": 1, "strang": 2}
FP_STRICT, FP_FAST = 0, 1
STATIC_MAX_STEPS = 4096   # CMFP_STATIC_MAX_STEPS
IPC_HANDLE_BYTES = 64     # CMFP_IPC_HANDLE_BYTES (sizeof(hipIpcMemHandle_t))


class Geometry(C.Structure):
    _fields_ = [
        ("dim", C.c_int),
        ("n_cell", C.c_int * MAX_DIM),
        ("prob_lo", C.c_double * MAX_DIM),
        ("dx", C.c_double * MAX_DIM),
        ("is_periodic", C.c_int * MAX_DIM),
        ("block_lo", C.c_int * MAX_DIM),
        ("block_n", C.c_int * MAX_DIM),
    ]


class Params(C.Structure):
    _fields_ = [
        ("lightspeed", C.c_double),
        ("Larmor", C.c_double),
        ("Debye", C.c_double),
        ("cfl", C.c_double),
        ("time_integrator", C.c_int),
        ("fp_mode", C.c_int),
        ("n0", C.c_double),
        ("force_dt", C.c_double),
    ]


class StateDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("reconstruction", C.c_int),
        ("flux", C.c_int),
        ("mass", C.c_double),
        ("charge", C.c_double),
        ("gamma", C.c_double),
        ("shock_threshold", C.c_double),
        ("div_transport", C.c_double),
        ("div_damping", C.c_double),
        ("effective_zero", C.c_double),
        ("viscosity", C.c_int),
        ("visc_mu0", C.c_double),
        ("visc_T0", C.c_double),
        ("visc_S", C.c_double),
        ("visc_n", C.c_double),
        ("visc_Pr", C.c_double),
        ("n_tracers", C.c_int),
        ("bc", ((C.c_int * 2) * MAX_DIM) * MAX_COMP),
        ("mixture", C.c_int),
        ("species_mass", C.c_double * (MAX_TRACERS + 1)),
        ("species_gamma", C.c_double * (MAX_TRACERS + 1)),
        ("species_charge", C.c_double * (MAX_TRACERS + 1)),
    ]


class ActionDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("corner_transport", C.c_int),
        ("solver", C.c_int),
        ("n_states", C.c_int),
        ("states", C.c_int * MAX_ACTION_STATES),
        ("cross_sections", (C.c_double * MAX_ACTION_STATES) * MAX_ACTION_STATES),
        ("species_component", C.c_int * MAX_ACTION_STATES),
    ]
