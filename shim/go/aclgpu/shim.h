/* Shared cgo preamble helpers: cgo cannot pass a Go function as a C callback directly, so the C ABI's callbacks are
 * routed through the exported trampolines of callbacks.go. */
#ifndef ACLGPU_SHIM_H
#define ACLGPU_SHIM_H
#include <stdlib.h>
#include "aclgpu.h"
extern void goReadCallback(void *user, acl_relationship_t *rel);
extern void goWatchCallback(void *user, uint64_t revision, int32_t op, acl_relationship_t *rel);
static inline int acl_read_go(acl_engine_t *h, const acl_filter_t *f, void *user) { return acl_read(h, f, (acl_read_cb)goReadCallback, user); }
static inline int acl_watch_poll_go(acl_engine_t *h, uint64_t after, const int *types, int n, void *user, uint64_t *rev) {
    return acl_watch_poll(h, after, types, n, user ? (acl_watch_cb)goWatchCallback : (acl_watch_cb)0, user, rev);
}
#endif
