/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See dlka_oracle_impl.h for scope, citations and the
 * "parity unpinned" statement.  Build: `make -C oracle` -> oracle/_build/libdlka_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SFX _f32
#include "dlka_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX _f64
#include "dlka_oracle_impl.h"
#undef REAL
#undef SFX

int dlka_oracle_abi_version(void) { return 1; }
