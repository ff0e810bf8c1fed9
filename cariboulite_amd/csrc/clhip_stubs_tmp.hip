// TEMPORARY: entry points not implemented yet (replaced later this round)
#include "clhip_common.h"
#define NI(name) clhip_set_error(name ": not implemented yet"); return -1
extern "C" int clhip_iir_cs16(const double *, int, double *, int16_t *, size_t, void *, size_t, void *) { NI("clhip_iir_cs16"); }
extern "C" size_t clhip_iir_workspace_bytes(size_t, int) { return 0; }
extern "C" clhip_tx_pipe *clhip_tx_pipe_create(int, double, double, const float *, int, int, int, int) { clhip_set_error("tx pipe: not implemented yet"); return nullptr; }
extern "C" void clhip_tx_pipe_destroy(clhip_tx_pipe *) {}
extern "C" void clhip_tx_pipe_reset(clhip_tx_pipe *) {}
extern "C" size_t clhip_tx_pipe_out_count(const clhip_tx_pipe *, size_t) { return 0; }
extern "C" long clhip_tx_pipe_run(clhip_tx_pipe *, int, const void *, size_t, size_t, uint8_t *, size_t, float *, size_t, void *) { NI("clhip_tx_pipe_run"); }
extern "C" int clhip_fm_demod(const float *, size_t, float *, float *, void *) { NI("clhip_fm_demod"); }
extern "C" int clhip_fm_mod(const float *, size_t, double, double, double *, float *, void *, size_t, void *) { NI("clhip_fm_mod"); }
extern "C" size_t clhip_fm_mod_workspace_bytes(size_t) { return 0; }
extern "C" int clhip_cw_tone(double, double, double, size_t, float *, void *) { NI("clhip_cw_tone"); }
