/*
 * dnsplat_constants.h — every magic number of the gsplat-1.0.0 rendering path
 * that dn-splatter calls (dn_splatter/dn_model.py:495-516, :564-575), in one
 * place.  Shared by the HIP kernels and by the CPU oracle so that a wrong
 * constant is wrong in exactly one spot (SURVEY.md §7 "Hard parts").
 * Values restated from SURVEY.md Appendix A (gsplat source is not vendored in
 * the reference; PARITY UNPINNED).
 */
#ifndef DNSPLAT_CONSTANTS_H
#define DNSPLAT_CONSTANTS_H

#define DNS_EPS2D_DEFAULT      0.3      /* A.2 step 4: screen-space blur added to cov2d diagonal   */
#define DNS_NEAR_DEFAULT       0.01     /* dn_model.py:507                                         */
#define DNS_FAR_DEFAULT        1e10     /* dn_model.py:508                                         */
#define DNS_FOV_CLAMP          1.3      /* A.2 step 3: tx,ty clamp at 1.3*tan(fov/2)               */
#define DNS_RADIUS_SIGMAS      3.0      /* A.2 step 6: radius = ceil(3*sqrt(lambda_max))           */
#define DNS_RADIUS_DISC_FLOOR  0.01     /* A.2 step 6: max(0.01, b^2-det) under the sqrt           */
#define DNS_ALPHA_MAX          0.999    /* A.6: alpha = min(0.999, o*exp(-sigma))                  */
#define DNS_ALPHA_MIN          (1.0 / 255.0) /* A.6: skip if alpha < 1/255                         */
#define DNS_T_MIN              1e-4     /* A.6: stop when T*(1-alpha) <= 1e-4                      */
#define DNS_ED_ALPHA_FLOOR     1e-10    /* A.6: expected depth = sum(w z)/max(alpha,1e-10)         */
#define DNS_TILE_DEFAULT       16       /* dn_model.py:470-472 BLOCK_WIDTH                         */
#define DNS_SH_C0              0.2820947917738781
#define DNS_MAX_CH             8        /* feature channels a splat record carries: 6 geometry floats + 8 channels + 2 spare = 16 floats */

#endif
