// inv_internal.h -- host-side helpers shared by the translation units of libinversus_hip.so (not part of the C ABI).
#pragma once

// Records the message inv_last_error() returns (thread-local) and hands `code` back: `return inv_fail(INV_ERR_X, "...")`.
int inv_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// CU count of the CURRENT device, cached per device ordinal (a process may drive several GPUs).
int inv_cu_count();
