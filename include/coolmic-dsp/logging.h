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

typedef enum coolmic_logging_level {
    COOLMIC_LOGGING_LEVEL_FATAL,
    COOLMIC_LOGGING_LEVEL_ERROR,
    COOLMIC_LOGGING_LEVEL_WARNING,
    COOLMIC_LOGGING_LEVEL_INFO,
    COOLMIC_LOGGING_LEVEL_DEBUG
} coolmic_logging_level_t;

const char *coolmic_logging_level2string(coolmic_logging_level_t level);

int coolmic_logging_log_real(const char *file, unsigned long int line, const char *component,
                             coolmic_logging_level_t level, int error, const char *format, ...)
    __attribute__((format(printf, 6, 7)));
#define coolmic_logging_log(level, error, ...) \
    coolmic_logging_log_real(__FILE__, __LINE__, COOLMIC_COMPONENT, (level), (error), __VA_ARGS__)

int coolmic_logging_set_cb_simple(int (*cb)(coolmic_logging_level_t level, const char *msg));

#ifdef __cplusplus
}
#endif
#endif
