/* Host-side exercise of the C ABI for the AddressSanitizer build (make -C sat-bundleadjust_amd/csrc asan_lib): argument checks,
 * error strings and handle lifetime -- the paths that run without a GPU (in a container without one every call that needs the
 * device must come back with SATBA_E_HIP and a message, never crash).  tests/test_host_logic.py builds and runs it.
 * ("-fsanitize=address host build of the C-ABI shim".) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "satba.h"

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) { fprintf(stderr, "abi_driver: %s failed (line %d): %s\n", #cond, __LINE__, satba_last_error()); return 1; } \
    } while (0)

int main(void) {
    satba_problem* p = (satba_problem*)0x1;
    CHECK(satba_version() >= 3);
    CHECK(satba_problem_create(NULL, &p) == SATBA_E_ARG);
    satba_problem_desc d;
    memset(&d, 0, sizeof d);
    d.cam_model = 7;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && p == NULL && strstr(satba_last_error(), "cam_model"));
    d.cam_model = SATBA_AFFINE; d.cam_param_len = 9;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && strstr(satba_last_error(), "cam_param_len"));
    d.cam_param_len = 8; d.n_params = 4;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && strstr(satba_last_error(), "n_params"));
    d.n_params = 5; d.n_cam = 0;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG);
    d.n_cam = 2; d.n_pts = 3; d.n_obs = 4; d.world = 1;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && strstr(satba_last_error(), "null input"));
    double cams[16] = {0}, obs[8] = {0}, w[4] = {1, 1, 1, 1};
    int32_t ci[4] = {0, 1, 0, 1}, pi[4] = {0, 0, 1, 1};
    d.cam_params = cams; d.cam_ind = ci; d.pts_ind = pi; d.pts2d = obs; d.weights = w;
    d.n_cam_fix = 5;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && strstr(satba_last_error(), "n_cam_fix"));
    d.n_cam_fix = 1; d.world = 2; d.rank = 2;
    CHECK(satba_problem_create(&d, &p) == SATBA_E_ARG && strstr(satba_last_error(), "rank"));
    d.world = 1; d.rank = 0;
    const int rc = satba_problem_create(&d, &p);
    if (rc == SATBA_OK) {  /* a GPU is present: a few calls on a live handle, then the teardown */
        double x[2 * 5 + 3 * 3] = {0};
        CHECK(satba_set_x(p, x) == SATBA_OK);
        CHECK(satba_set_x(p, NULL) == SATBA_E_ARG);
        CHECK(satba_prepare(p, 1) == SATBA_E_STATE);
        double err[4], cost = -1.0;
        CHECK(satba_reprojection_errors(p, NULL, NULL) == SATBA_E_ARG);
        CHECK(satba_reprojection_errors(p, err, &cost) == SATBA_OK && cost >= 0.0);
        double err2[4] = {-1.0, -1.0, -1.0, -1.0};
        CHECK(satba_reprojection_errors_fetch(p, err2) == SATBA_E_STATE);  /* no _begin yet */
        CHECK(satba_reprojection_errors_begin(p) == SATBA_OK && satba_reprojection_errors_fetch(p, NULL) == SATBA_E_ARG);
        CHECK(satba_reprojection_errors_fetch(p, err2) == SATBA_OK && memcmp(err, err2, sizeof err) == 0);
        CHECK(satba_configure(p, 9, 1.0) == SATBA_E_ARG && satba_configure(p, 1, -1.0) == SATBA_E_ARG);
        double out[16];
        CHECK(satba_get_info(p, out, 4) == SATBA_E_ARG && satba_get_info(p, out, 16) == SATBA_OK);
        satba_problem_destroy(p);
    } else {
        CHECK(rc == SATBA_E_HIP && p == NULL && strlen(satba_last_error()) > 0);
    }
    /* entry points that take no handle */
    CHECK(satba_set_x(NULL, NULL) == SATBA_E_ARG && satba_linearize(NULL) == SATBA_E_ARG && satba_accept(NULL) == SATBA_E_ARG);
    CHECK(satba_header_len(NULL) == 0 && satba_exchange_len(NULL) == 0 && satba_layout_len(NULL, 0) == -1);
    CHECK(satba_lm_run(NULL, 1, 0, 0.0, NULL, 0) == SATBA_E_ARG && satba_lm_state(NULL, NULL, 0) == SATBA_E_ARG);
    CHECK(satba_reprojection_errors(NULL, NULL, NULL) == SATBA_E_ARG);
    CHECK(satba_reprojection_errors_begin(NULL) == SATBA_E_ARG && satba_reprojection_errors_fetch(NULL, NULL) == SATBA_E_ARG);
    CHECK(satba_rpc_fit(-1, 0, NULL, NULL, 1e-3, 1e-2, 20, NULL, NULL, NULL, 0) == SATBA_E_ARG);
    CHECK(satba_rpc_refit(1, NULL, NULL, NULL, NULL, NULL, 10, 1e-3, 1e-2, 20, NULL, NULL, NULL, NULL, NULL, 0) == SATBA_E_ARG);
    CHECK(satba_rpc_localization(NULL, 3, NULL, NULL, NULL, NULL, NULL, 0) == SATBA_E_ARG);
    CHECK(satba_triangulate_pairwise(SATBA_RPC, NULL, NULL, 1, NULL, NULL, NULL, NULL, 0, NULL) == SATBA_E_ARG);
    CHECK(satba_init_pts3d(SATBA_AFFINE, 2, 1, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, 0, 1, NULL) == SATBA_E_ARG);
    satba_problem_destroy(NULL);
    printf("abi_driver ok\n");
    return 0;
}
