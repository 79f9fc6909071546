// phf_hier3_isa.hip — host side of the hand-allocated gfx950 code object (generated/phf_hier3_gfx950.s, emitted by
// tools/gen_hier_isa.py and assembled by pyhillfit_amd/build.py): the code object travels INSIDE libpyhillfit_amd.so (.incbin), is
// loaded once per device with hipModuleLoadData, and its kernels — the hierarchical iteration per (experiments, point shape):
// generated/phf_hier3_isa_layout.h phf_isa_hier_kernels[]; phf_hier_fused_advance, one persistent grid with a body per entry of that
// table; phf_sl3_advance; the unit kernels — are launched with hipModuleLaunchKernel on the caller's stream.
// The constants blob the kernels read — the exp2 / log / erfc / normal tables of phf_math.h followed by the scalar constants of
// generated/phf_hier3_isa_layout.h — is built here from the SAME arrays the hipcc kernels and the host twin compile.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/pyhillfit_amd.h"
#include "phf_common.h"
#include "phf_hier3_isa.h"
#include "phf_math.h"

#if !defined(__HIP_DEVICE_COMPILE__)
#ifndef PHF_ISA_CO_PATH
#error "PHF_ISA_CO_PATH: path of the assembled code object (pyhillfit_amd/build.py passes it)"
#endif
asm(".section .rodata\n"
    ".balign 4096\n"
    ".global phf_isa_code_object\n"
    "phf_isa_code_object:\n"
    ".incbin \"" PHF_ISA_CO_PATH "\"\n"
    ".global phf_isa_code_object_end\n"
    "phf_isa_code_object_end:\n"
    ".byte 0\n"
    ".previous\n");
#endif
extern "C" const char phf_isa_code_object[];          // (declared for the device pass too: host functions are parsed there)
extern "C" const char phf_isa_code_object_end[];

namespace {

constexpr int kMaxDevices = 64;
const char* const kUnitNames[] = {"phf_isa_unit_exp_fast", "phf_isa_unit_exp_capped", "phf_isa_unit_log_pos", "phf_isa_unit_log_fast",
                                  "phf_isa_unit_erfc_tab", "phf_isa_unit_rcp",        "phf_isa_unit_sqrt_nonneg", "phf_isa_unit_normal_u32",
                                  "phf_isa_unit_log_u",    "phf_isa_unit_philox7"};
constexpr int kNumUnits = sizeof(kUnitNames) / sizeof(kUnitNames[0]);

struct DeviceModule {
  std::once_flag once;
  int rc = PHF_ERR_HIP;
  char msg[256] = "";
  hipModule_t module = nullptr;
  hipFunction_t advance[PHF_ISA_HIER_NUM_KERNELS] = {};      // phf_isa_hier_kernels[]: one per (experiments, point shape)
  hipFunction_t fused = nullptr;                               // phf_hier_fused_advance: every body in one persistent grid
  hipFunction_t sl_advance = nullptr;
  hipFunction_t unit[kNumUnits] = {};
  void* consts = nullptr;                      // tables + scalar constants, device memory, lives as long as the process
};
DeviceModule g_modules[kMaxDevices];

std::vector<unsigned char> build_blob() {
  static_assert(PHF_ISA_LOGPHI_BLOB_OFF >= PHF_ISA_CONST_OFF + 8 * PHF_ISA_NUM_CONSTS && PHF_ISA_LOGPHI_BLOB_OFF % 16 == 0, "blob layout");
  std::vector<unsigned char> b(PHF_ISA_LOGPHI_BLOB_OFF + sizeof(phf_t_logphi) + 4096, 0);     // (+ slack: the staging loads whole rounds)
  std::memcpy(b.data() + PHF_ISA_LOGPHI_BLOB_OFF, phf_t_logphi, sizeof(phf_t_logphi));
  static_assert(sizeof(phf_t_exp2) == 512 && sizeof(phf_t_log) == PHF_LOG_TAB_N * 16, "table sizes");
  static_assert(PHF_ISA_LOG_OFF == 512 && PHF_ISA_ERFC_OFF == PHF_ISA_LOG_OFF + PHF_LOG_TAB_N * 16, "LDS image of the tables");
  static_assert(PHF_ISA_NORMAL_OFF == PHF_ISA_ERFC_OFF + PHF_ERFC_TAB_N * 96 && PHF_ISA_TABLE_BYTES == PHF_ISA_NORMAL_OFF + PHF_NORMAL_TAB_N * 48,
                "LDS image of the tables");
  std::memcpy(b.data() + PHF_ISA_EXP2_OFF, phf_t_exp2, sizeof(phf_t_exp2));
  std::memcpy(b.data() + PHF_ISA_LOG_OFF, phf_t_log, sizeof(phf_t_log));
  std::memcpy(b.data() + PHF_ISA_ERFC_OFF, phf_t_erfc, sizeof(phf_t_erfc));
  std::memcpy(b.data() + PHF_ISA_NORMAL_OFF, phf_t_normal, sizeof(phf_t_normal));
  std::memcpy(b.data() + PHF_ISA_CONST_OFF, phf_isa_const_bits, sizeof(phf_isa_const_bits));
  // the polynomial coefficients in the generated list must be the header's (a change of phf_math.h without regenerating would
  // otherwise go unnoticed until a parity test): checked once, here
  return b;
}

bool coefficients_match() {
  // order of phf_isa_const_bits: tools/isa/phf_isa_math.py CONSTS
  const double want[] = {PHF_EXP_MAGIC, PHF_64_LOG2E, -PHF_LN2_64_HI, -PHF_LN2_64_LO, phf_k_exp[0], phf_k_exp[1], phf_k_exp[2], phf_k_exp[3],
                         phf_k_log[0], phf_k_log[1], phf_k_log[2], phf_k_log[3], PHF_LN2_HI, PHF_LN2_LO, PHF_LN10, PHF_INV_SQRT2};
  for (size_t i = 0; i < sizeof(want) / sizeof(want[0]); ++i) {
    uint64_t u;
    std::memcpy(&u, &want[i], 8);
    if (u != phf_isa_const_bits[i]) return false;
  }
  return true;
}

void load_module(DeviceModule* m) {
  auto fail = [m](const char* what, hipError_t e) {
    std::snprintf(m->msg, sizeof(m->msg), "gfx950 code object: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    m->rc = PHF_ERR_HIP;
  };
  if (!coefficients_match()) {
    std::snprintf(m->msg, sizeof(m->msg), "gfx950 code object: constants of generated/phf_hier3_isa_layout.h differ from phf_math.h (re-run tools/gen_hier_isa.py)");
    m->rc = PHF_ERR_UNSUPPORTED;
    return;
  }
  hipError_t e = hipModuleLoadData(&m->module, phf_isa_code_object);
  if (e != hipSuccess) return fail("hipModuleLoadData", e);
  for (int i = 0; i < kNumUnits; ++i) {
    e = hipModuleGetFunction(&m->unit[i], m->module, kUnitNames[i]);
    if (e != hipSuccess) return fail(kUnitNames[i], e);
  }
  for (int i = 0; i < PHF_ISA_HIER_NUM_KERNELS; ++i) {
    e = hipModuleGetFunction(&m->advance[i], m->module, phf_isa_hier_kernels[i].name);
    if (e != hipSuccess) { m->advance[i] = nullptr; (void)hipGetLastError(); }   // a units-only code object (generator bring-up)
  }
  e = hipModuleGetFunction(&m->fused, m->module, "phf_hier_fused_advance");
  if (e != hipSuccess) { m->fused = nullptr; (void)hipGetLastError(); }
  e = hipModuleGetFunction(&m->sl_advance, m->module, "phf_sl3_advance");
  if (e != hipSuccess) { m->sl_advance = nullptr; (void)hipGetLastError(); }
  const std::vector<unsigned char> blob = build_blob();
  e = hipMalloc(&m->consts, blob.size());
  if (e != hipSuccess) return fail("hipMalloc(constants)", e);
  e = hipMemcpy(m->consts, blob.data(), blob.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail("hipMemcpy(constants)", e);
  m->rc = PHF_OK;
}

int get_module(DeviceModule** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  if (dev < 0 || dev >= kMaxDevices) return phf_fail(PHF_ERR_UNSUPPORTED, "gfx950 code object: device index out of range");
  DeviceModule* m = &g_modules[dev];
  std::call_once(m->once, load_module, m);
  if (m->rc != PHF_OK) return phf_fail(m->rc, m->msg);
  *out = m;
  return PHF_OK;
}

int launch(hipFunction_t f, unsigned blocks, void* args, size_t bytes, hipStream_t stream, const char* what) {
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, extra);
  if (e != hipSuccess) {
    std::snprintf(phf_error_buffer(), kPhfErrorBufferSize, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return PHF_ERR_HIP;
  }
  return PHF_OK;
}

}  // namespace

int phf_hier_isa_find(int n_expts, int shape_code) {
  int which = -1;
  for (int i = 0; i < PHF_ISA_HIER_NUM_KERNELS; ++i)
    if (phf_isa_hier_kernels[i].n_expts == n_expts && phf_isa_hier_kernels[i].shape_code == shape_code) which = i;
  if (which < 0) return -1;                                  // (before the module is touched: most launches have no such kernel)
  DeviceModule* m = nullptr;
  if (get_module(&m) != PHF_OK) return -1;
  return m->advance[which] ? which : -1;
}

int phf_hier_isa_scratch_slots(int n_expts, int shape_code) {
  for (int i = 0; i < PHF_ISA_HIER_NUM_KERNELS; ++i)
    if (phf_isa_hier_kernels[i].n_expts == n_expts && phf_isa_hier_kernels[i].shape_code == shape_code) return phf_isa_hier_kernels[i].scratch_slots;
  return 0;
}

int phf_hier_isa_advance(int which, phf_hier3_isa_args* a, int grid_waves, hipStream_t stream) {
  DeviceModule* m = nullptr;
  if (int rc = get_module(&m)) return rc;
  if (which < 0 || which >= PHF_ISA_HIER_NUM_KERNELS || !m->advance[which])
    return phf_fail(PHF_ERR_UNSUPPORTED, "gfx950 code object holds no such hierarchical kernel");
  a->consts = m->consts;
  const unsigned blocks = (unsigned)((grid_waves + 3) / 4);
  return launch(m->advance[which], blocks, a, sizeof(*a), stream, "phf_hierarchical_advance (gfx950 assembly)");
}

int phf_hier_isa_fused_advance(phf_hier_fused_args* a, int grid_waves, hipStream_t stream) {
  DeviceModule* m = nullptr;
  if (int rc = get_module(&m)) return rc;
  if (!m->fused) return phf_fail(PHF_ERR_UNSUPPORTED, "gfx950 code object holds no phf_hier_fused_advance");
  a->consts = m->consts;
  return launch(m->fused, (unsigned)((grid_waves + 3) / 4), a, sizeof(*a), stream, "phf_hierarchical_advance_fused (gfx950 assembly)");
}

bool phf_sl3_isa_available() {
  DeviceModule* m = nullptr;
  if (get_module(&m) != PHF_OK) return false;
  return m->sl_advance != nullptr;
}

int phf_sl3_isa_advance(phf_sl3_isa_args* a, int grid_waves, hipStream_t stream) {
  DeviceModule* m = nullptr;
  if (int rc = get_module(&m)) return rc;
  if (!m->sl_advance) return phf_fail(PHF_ERR_UNSUPPORTED, "gfx950 code object holds no phf_sl3_advance");
  a->consts = m->consts;
  return launch(m->sl_advance, (unsigned)((grid_waves + 3) / 4), a, sizeof(*a), stream, "phf_single_level_advance (gfx950 assembly, model 2)");
}

extern "C" int phf_debug_isa(int fn, int64_t n, const void* in, void* out, void* stream) {
  if (fn < 0 || fn >= kNumUnits) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_debug_isa: unknown function");
  if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!in || !out))) return phf_fail(PHF_ERR_INVALID_ARGUMENT, "phf_debug_isa: bad arguments");
  if (n == 0) return PHF_OK;
  DeviceModule* m = nullptr;
  if (int rc = get_module(&m)) return rc;
  struct { const void* consts; const void* in; void* out; uint32_t n; uint32_t pad; } args = {m->consts, in, out, (uint32_t)n, 0u};
  return launch(m->unit[fn], (unsigned)((n + 255) / 256), &args, sizeof(args), (hipStream_t)stream, "phf_debug_isa");
}
