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

/* codec names of the encoder stage (not built here; a host's sources name them: ref: coolmic-dsp.h:32-33) */
#define COOLMIC_DSP_CODEC_VORBIS "audio/ogg; codec=vorbis"
#define COOLMIC_DSP_CODEC_OPUS   "audio/ogg; codec=opus"

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

/* feature names (ref: coolmic-dsp.h:53-58).  Encoders and hardware drivers are the host's: inside the reference's
 * build their tokens come from its own flags (make dropin HOST_FEATURES=...); the stand-alone library reports the
 * drivers it brings.  driver:sine and accel:* are this library's additions. */
#define COOLMIC_FEATURE_ENCODE_OGG_VORBIS   "encode:ogg/vorbis"
#define COOLMIC_FEATURE_ENCODE_OGG_OPUS     "encode:ogg/opus"
#define COOLMIC_FEATURE_DRIVER_NULL         "driver:null"
#define COOLMIC_FEATURE_DRIVER_OSS          "driver:oss"
#define COOLMIC_FEATURE_DRIVER_OPENSL       "driver:opensl"
#define COOLMIC_FEATURE_DRIVER_STDIO        "driver:stdio"
#define COOLMIC_FEATURE_DRIVER_SINE         "driver:sine"
#define COOLMIC_FEATURE_ACCEL_HIP           "accel:hip/gfx950"

/* static text for an error number (ref: src/coolmic-dsp.c, coolmic_error2string) */
const char *coolmic_error2string(const int error);
/* space separated feature list; coolmic_feature_check(): 1 yes, 0 no, <0 error */
const char *coolmic_features(void);
int coolmic_feature_check(const char *feature);

#ifdef __cplusplus
}
#endif
#endif
