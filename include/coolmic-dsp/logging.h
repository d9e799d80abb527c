/*
 * logging.h -- the library-wide log callback (ref: include/coolmic-dsp/logging.h:31-47,
 * src/logging.c:59-107).  One global callback receives fully formatted lines:
 *   "<component> in <file>:<line>: <LEVEL>: <message>[: <error text>]"
 * With no callback installed logging costs one load and a return.
 */
#ifndef __COOLMIC_DSP_LOGGING_H__
#define __COOLMIC_DSP_LOGGING_H__

#ifdef __cplusplus
extern "C" {
#endif

/* Severity of a line, most severe first.  The stages of the hot path log per read at DEBUG
 * (ref: src/vumeter.c:119-154) and report a missing GPU or a failed launch at ERROR. */
typedef enum coolmic_logging_level {
    COOLMIC_LOGGING_LEVEL_FATAL,
    COOLMIC_LOGGING_LEVEL_ERROR,
    COOLMIC_LOGGING_LEVEL_WARNING,
    COOLMIC_LOGGING_LEVEL_INFO,
    COOLMIC_LOGGING_LEVEL_DEBUG
} coolmic_logging_level_t;

/* "FATAL", "ERROR", ... as they appear in a formatted line; "(unknown)" for anything else */
const char *coolmic_logging_level2string(coolmic_logging_level_t level);

/* Formats one line and hands it to the callback.  `error` is a COOLMIC_ERROR_* code whose text is
 * appended, or COOLMIC_ERROR_NONE.  Callable from any thread; returns COOLMIC_ERROR_NONE, also when
 * no callback is installed (COOLMIC_ERROR_FAULT without a format, COOLMIC_ERROR_NOMEM if a line cannot
 * be built).  Use the coolmic_logging_log() macro: it supplies file, line and the
 * COOLMIC_COMPONENT string that every translation unit defines before including this header. */
int coolmic_logging_log_real(const char *file, unsigned long int line, const char *component,
                             coolmic_logging_level_t level, int error, const char *format, ...)
    __attribute__((format(printf, 6, 7)));
#define coolmic_logging_log(level, error, ...) \
    coolmic_logging_log_real(__FILE__, __LINE__, COOLMIC_COMPONENT, (level), (error), __VA_ARGS__)

/* Installs (or with NULL removes) the one process-wide callback.  It receives the level and the
 * formatted line, must not log itself, and may be called from any thread that logs. */
int coolmic_logging_set_cb_simple(int (*cb)(coolmic_logging_level_t level, const char *msg));

#ifdef __cplusplus
}
#endif
#endif
