/*
 * coolmic-dsp.h -- error numbers and the feature string of the MI355X build.
 *
 * The error numbers are part of the drop-in boundary and equal the reference's
 * (ref: include/coolmic-dsp/coolmic-dsp.h:36-50).
 */
#ifndef __COOLMIC_DSP_COOLMIC_DSP_H__
#define __COOLMIC_DSP_COOLMIC_DSP_H__

#ifdef __cplusplus
extern "C" {
#endif

#define COOLMIC_ERROR_NONE              (  0)
#define COOLMIC_ERROR_GENERIC           ( -1)
#define COOLMIC_ERROR_NOSYS             ( -8)
#define COOLMIC_ERROR_FAULT             ( -9)
#define COOLMIC_ERROR_INVAL             (-10)
#define COOLMIC_ERROR_NOMEM             (-11)
#define COOLMIC_ERROR_BUSY              (-12)
#define COOLMIC_ERROR_PERM              (-13)
#define COOLMIC_ERROR_CONNREFUSED       (-14)
#define COOLMIC_ERROR_CONNECTED         (-15)
#define COOLMIC_ERROR_UNCONNECTED       (-16)
#define COOLMIC_ERROR_NOTLS             (-17)
#define COOLMIC_ERROR_TLSBADCERT        (-18)
#define COOLMIC_ERROR_BADRQC            (-19)
#define COOLMIC_ERROR_RETRY             (-20)

/* feature names this build can report (ref: coolmic-dsp.h:53-58 for the driver ones) */
#define COOLMIC_FEATURE_DRIVER_NULL     "driver:null"
#define COOLMIC_FEATURE_DRIVER_SINE     "driver:sine"
#define COOLMIC_FEATURE_DRIVER_STDIO    "driver:stdio"
#define COOLMIC_FEATURE_ACCEL_HIP       "accel:hip/gfx950"

/* static text for an error number (ref: src/coolmic-dsp.c, coolmic_error2string) */
const char *coolmic_error2string(const int error);
/* space separated feature list; coolmic_feature_check(): 1 yes, 0 no, <0 error */
const char *coolmic_features(void);
int coolmic_feature_check(const char *feature);

#ifdef __cplusplus
}
#endif
#endif
