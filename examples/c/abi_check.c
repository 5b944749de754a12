/*
 * The smallest C consumer of libgpd.so: include/gpd.h is plain C (C11, -pedantic clean), and a binding that mirrors its structs
 * verifies them against the library before its first call (what gym_pybullet_drones_amd/_native.py does through ctypes,
 * INTEGRATION.md section 2).  Needs no GPU: nothing is launched.
 *
 *   gcc -std=c11 -I include examples/c/abi_check.c -L gym_pybullet_drones_amd/csrc -lgpd \
 *       -Wl,-rpath,$PWD/gym_pybullet_drones_amd/csrc -o abi_check && ./abi_check
 */
#include <stdio.h>

#include "gpd.h"

int main(void) {
    int32_t sizes[3];
    gpd_struct_sizes(sizes);
    printf("libgpd ABI %d (header %d); GpdParams %d/%zu GpdState %d/%zu GpdStepCfg %d/%zu GpdSwarm %d/%zu bytes (library/header)\n",
           gpd_abi_version(), GPD_ABI_VERSION, sizes[0], sizeof(GpdParams), sizes[1], sizeof(GpdState), sizes[2], sizeof(GpdStepCfg),
           gpd_sizeof_swarm(), sizeof(GpdSwarm));
    /* argument errors come back as codes with a message, never as a crash */
    const int rc = gpd_step(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    printf("gpd_step(NULL, ...) -> %d: %s\n", rc, gpd_last_error());
    return !(gpd_abi_version() == GPD_ABI_VERSION && sizes[0] == (int)sizeof(GpdParams) && sizes[1] == (int)sizeof(GpdState) &&
             sizes[2] == (int)sizeof(GpdStepCfg) && gpd_sizeof_swarm() == (int)sizeof(GpdSwarm) && rc == GPD_EINVAL);
}
