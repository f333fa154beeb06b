// layout_hooks.cpp -- part of libmipt_diag.so (test infrastructure): C entry points to the host restatements of the device layout's two
// orders (tests/cpp/layout_order.cpp: mipt::pair_order / tri_slots; pair_order_top is the product's constant) so that
// tests/test_host_layout.py and tests/tools/layout_model.py can call them.
#include "../../include/mipt_diag.h"
#include "../../rust_ray_tracing_amd/csrc/mipt_internal.h"


extern "C" {
int mipt_internal_pair_order(const void *nodes, uint32_t n_nodes, uint32_t *order_out, uint32_t cap, uint32_t *n_records_out) {
    return mipt::pair_order((const MiptNode *)nodes, n_nodes, order_out, cap, n_records_out);
}
uint32_t mipt_internal_pair_order_top(void) { return mipt::pair_order_top(); }
int mipt_internal_tri_slots(const void *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *slot_out, uint32_t *n_slots_out) {
    return mipt::tri_slots((const MiptNode *)nodes, n_nodes, n_tris, slot_out, n_slots_out);
}
}
