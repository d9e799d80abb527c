/* logging.c -- one global log callback (contract: <coolmic-dsp/logging.h>;
 * ref: src/logging.c:34-107). */
#include "host_internal.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int (*log_cb_t)(coolmic_logging_level_t level, const char *msg);

static pthread_mutex_t cb_lock = PTHREAD_MUTEX_INITIALIZER;
static log_cb_t cb_simple;

const char *coolmic_logging_level2string(coolmic_logging_level_t level)
{
    static const char *const names[] = {"FATAL", "ERROR", "WARNING", "INFO", "DEBUG"};
    if ((unsigned int)level < sizeof(names) / sizeof(names[0]))
        return names[level];
    return "(unknown)";
}

int coolmic_logging_log_real(const char *file, unsigned long int line, const char *component,
                             coolmic_logging_level_t level, int error, const char *format, ...)
{
    log_cb_t cb = __atomic_load_n(&cb_simple, __ATOMIC_ACQUIRE);
    char *user = NULL, *full = NULL;
    va_list ap;
    int n;

    if (format == NULL)
        return COOLMIC_ERROR_FAULT;
    if (cb == NULL)                          /* the hot path logs at DEBUG on every read */
        return COOLMIC_ERROR_NONE;
    /* a source path is shown from its "coolmic/" directory on (ref: src/logging.c:66-68) */
    if (file != NULL) {
        const char *cut = strstr(file, "/coolmic/");
        if (cut != NULL)
            file = cut + 1;
    }

    va_start(ap, format);
    n = vasprintf(&user, format, ap);
    va_end(ap);
    if (n < 0)
        return COOLMIC_ERROR_NOMEM;

    if (error == COOLMIC_ERROR_NONE)
        n = asprintf(&full, "%s in %s:%lu: %s: %s", component, file, line,
                     coolmic_logging_level2string(level), user);
    else
        n = asprintf(&full, "%s in %s:%lu: %s: %s: %s", component, file, line,
                     coolmic_logging_level2string(level), user, coolmic_error2string(error));
    free(user);
    if (n < 0)
        return COOLMIC_ERROR_NOMEM;
    cb(level, full);
    free(full);
    return COOLMIC_ERROR_NONE;
}

int coolmic_logging_set_cb_simple(int (*cb)(coolmic_logging_level_t level, const char *msg))
{
    pthread_mutex_lock(&cb_lock);
    __atomic_store_n(&cb_simple, cb, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&cb_lock);
    return COOLMIC_ERROR_NONE;
}
