/* sz_conf.c -- sz.config loader of the MI355X SZ build (host C).
 *
 * Same keys, defaults and failure behaviour as the reference's SZ_ReadConf / SZ_LoadConf
 * (sz/src/conf.c:74-391, :403-...), with a small self-contained INI reader instead of the
 * bundled iniparser: sections [ENV] and [PARAMETER], `key = value`, `#`/`;` comments,
 * case-insensitive section:key lookup. */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "sz.h"

typedef struct { char *key; char *val; } ini_kv;
typedef struct { ini_kv *kv; size_t n, cap; } ini_t;

static char *trim(char *s)
{
    while (*s && isspace((unsigned char)*s)) s++;
    char *e = s + strlen(s);
    while (e > s && isspace((unsigned char)e[-1])) *--e = 0;
    return s;
}

static void ini_free(ini_t *d)
{
    for (size_t i = 0; i < d->n; i++) { free(d->kv[i].key); free(d->kv[i].val); }
    free(d->kv);
    d->kv = NULL; d->n = d->cap = 0;
}

static int ini_load(const char *path, ini_t *d)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[2048], section[256] = "";
    memset(d, 0, sizeof(*d));
    while (fgets(line, sizeof(line), f)) {
        char *s = trim(line);
        if (!*s || *s == '#' || *s == ';') continue;
        if (*s == '[') {
            char *e = strchr(s, ']');
            if (!e) continue;
            *e = 0;
            snprintf(section, sizeof(section), "%s", trim(s + 1));
            for (char *p = section; *p; p++) *p = (char)tolower((unsigned char)*p);
            continue;
        }
        char *eq = strchr(s, '=');
        if (!eq) continue;
        *eq = 0;
        char *k = trim(s), *v = trim(eq + 1);
        /* strip a trailing comment */
        for (char *p = v; *p; p++) if ((*p == '#' || *p == ';') && (p == v || isspace((unsigned char)p[-1]))) { *p = 0; break; }
        v = trim(v);
        if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 32; d->kv = (ini_kv *)realloc(d->kv, d->cap * sizeof(ini_kv)); }
        size_t kl = strlen(section) + 1 + strlen(k) + 1;
        d->kv[d->n].key = (char *)malloc(kl);
        snprintf(d->kv[d->n].key, kl, "%s:%s", section, k);
        for (char *p = d->kv[d->n].key; *p; p++) *p = (char)tolower((unsigned char)*p);
        d->kv[d->n].val = strdup(v);
        d->n++;
    }
    fclose(f);
    return 0;
}

static const char *ini_str(const ini_t *d, const char *key, const char *def)
{
    char low[256];
    snprintf(low, sizeof(low), "%s", key);
    for (char *p = low; *p; p++) *p = (char)tolower((unsigned char)*p);
    for (size_t i = 0; i < d->n; i++) if (strcmp(d->kv[i].key, low) == 0) return d->kv[i].val;
    return def;
}
static long ini_int(const ini_t *d, const char *key, long def)
{
    const char *s = ini_str(d, key, NULL);
    return s ? strtol(s, NULL, 0) : def;
}
static double ini_dbl(const ini_t *d, const char *key, double def)
{
    const char *s = ini_str(d, key, NULL);
    return s ? atof(s) : def;
}

static void update_quant_info(int q) { exe_params->intvCapacity = q; exe_params->intvRadius = q / 2; }

int SZ_ReadConf(const char *sz_cfgFile)
{
    confparams_cpr = (sz_params *)calloc(1, sizeof(sz_params));
    exe_params = (sz_exedata *)calloc(1, sizeof(sz_exedata));
    sysEndianType = LITTLE_ENDIAN_SYSTEM;
    confparams_cpr->plus_bits = 3;

    if (sz_cfgFile == NULL) { /* conf.c:97-141 */
        dataEndianType = LITTLE_ENDIAN_DATA;
        confparams_cpr->sol_ID = SZ;
        confparams_cpr->max_quant_intervals = 65536;
        confparams_cpr->maxRangeRadius = confparams_cpr->max_quant_intervals / 2;
        exe_params->intvCapacity = confparams_cpr->maxRangeRadius * 2;
        exe_params->intvRadius = confparams_cpr->maxRangeRadius;
        confparams_cpr->quantization_intervals = 0;
        exe_params->optQuantMode = 1;
        confparams_cpr->predThreshold = 0.99f;
        confparams_cpr->sampleDistance = 100;
        confparams_cpr->szMode = SZ_BEST_COMPRESSION;
        confparams_cpr->losslessCompressor = ZSTD_COMPRESSOR;
        confparams_cpr->gzipMode = 3;
        confparams_cpr->errorBoundMode = PSNR;
        confparams_cpr->psnr = 90;
        confparams_cpr->absErrBound = 1E-4;
        confparams_cpr->relBoundRatio = 1E-4;
        confparams_cpr->accelerate_pw_rel_compression = 1;
        confparams_cpr->pw_relBoundRatio = 1E-3;
        confparams_cpr->segment_size = 36;
        confparams_cpr->pwr_type = SZ_PWR_MIN_TYPE;
        confparams_cpr->snapshotCmprStep = 5;
        confparams_cpr->withRegression = SZ_WITH_LINEAR_REGRESSION;
        confparams_cpr->randomAccess = 0;
        confparams_cpr->protectValueRange = 0;
        return SZ_SCES;
    }

    if (access(sz_cfgFile, F_OK) != 0) { printf("[SZ] Configuration file NOT accessible.\n"); return SZ_NSCS; }
    ini_t ini;
    if (ini_load(sz_cfgFile, &ini) != 0) { printf("[SZ] Iniparser failed to parse the conf. file.\n"); return SZ_NSCS; }

#define CONF_FAIL(msg) do { printf("%s\n", msg); ini_free(&ini); return SZ_NSCS; } while (0)
    const char *s = ini_str(&ini, "ENV:dataEndianType", "LITTLE_ENDIAN_DATA");
    if (strcmp(s, "LITTLE_ENDIAN_DATA") == 0) dataEndianType = LITTLE_ENDIAN_DATA;
    else if (strcmp(s, "BIG_ENDIAN_DATA") == 0) dataEndianType = BIG_ENDIAN_DATA;
    else CONF_FAIL("Error: Wrong dataEndianType: please set it correctly in sz.config.");

    s = ini_str(&ini, "ENV:sol_name", "");
    if (strcmp(s, "SZ") == 0) confparams_cpr->sol_ID = SZ;
    else if (strcmp(s, "SZ_Transpose") == 0) confparams_cpr->sol_ID = SZ_Transpose;
    else if (strcmp(s, "PASTRI") == 0) CONF_FAIL("[SZ] Error: the PASTRI solution is outside the scope of the MI355X build");
    else CONF_FAIL("[SZ] Error: wrong solution name (please check sz.config file)");

    int max_quant_intervals = (int)ini_int(&ini, "PARAMETER:max_quant_intervals", 65536);
    confparams_cpr->max_quant_intervals = (unsigned)max_quant_intervals;
    int quantization_intervals = (int)ini_int(&ini, "PARAMETER:quantization_intervals", 0);
    confparams_cpr->quantization_intervals = (unsigned)quantization_intervals;
    if (quantization_intervals > 0) {
        update_quant_info(quantization_intervals);
        confparams_cpr->max_quant_intervals = (unsigned)quantization_intervals;
        exe_params->optQuantMode = 0;
    } else {
        confparams_cpr->maxRangeRadius = (unsigned)max_quant_intervals / 2;
        exe_params->intvCapacity = confparams_cpr->maxRangeRadius * 2;
        exe_params->intvRadius = confparams_cpr->maxRangeRadius;
        exe_params->optQuantMode = 1;
    }
    if (quantization_intervals % 2 != 0) CONF_FAIL("Error: quantization_intervals must be an even number!");

    confparams_cpr->predThreshold = (float)ini_dbl(&ini, "PARAMETER:predThreshold", 0);
    confparams_cpr->sampleDistance = (int)ini_int(&ini, "PARAMETER:sampleDistance", 0);

    s = ini_str(&ini, "PARAMETER:szMode", NULL);
    if (!s) CONF_FAIL("[SZ] Error: Null szMode setting (please check sz.config file)");
    if (strcmp(s, "SZ_BEST_SPEED") == 0) confparams_cpr->szMode = SZ_BEST_SPEED;
    else if (strcmp(s, "SZ_DEFAULT_COMPRESSION") == 0) confparams_cpr->szMode = SZ_DEFAULT_COMPRESSION;
    else if (strcmp(s, "SZ_BEST_COMPRESSION") == 0) confparams_cpr->szMode = SZ_BEST_COMPRESSION;
    else CONF_FAIL("[SZ] Error: Wrong szMode setting (please check sz.config file)");

    s = ini_str(&ini, "PARAMETER:losslessCompressor", "ZSTD_COMPRESSOR");
    if (strcmp(s, "GZIP_COMPRESSOR") == 0) confparams_cpr->losslessCompressor = GZIP_COMPRESSOR;
    else if (strcmp(s, "ZSTD_COMPRESSOR") == 0) confparams_cpr->losslessCompressor = ZSTD_COMPRESSOR;
    else CONF_FAIL("[SZ] Error: Wrong losslessCompressor setting (please check sz.config file)");

    s = ini_str(&ini, "PARAMETER:withLinearRegression", "YES");
    confparams_cpr->withRegression = (strcmp(s, "YES") == 0 || strcmp(s, "yes") == 0) ? SZ_WITH_LINEAR_REGRESSION : SZ_NO_REGRESSION;

    s = ini_str(&ini, "PARAMETER:gzipMode", "Gzip_BEST_SPEED");
    if (strcmp(s, "Gzip_NO_COMPRESSION") == 0) confparams_cpr->gzipMode = 0;
    else if (strcmp(s, "Gzip_BEST_SPEED") == 0) confparams_cpr->gzipMode = 1;
    else if (strcmp(s, "Gzip_BEST_COMPRESSION") == 0) confparams_cpr->gzipMode = 9;
    else if (strcmp(s, "Gzip_DEFAULT_COMPRESSION") == 0) confparams_cpr->gzipMode = -1;
    else CONF_FAIL("[SZ] Error: Wrong gzip Mode (please check sz.config file)");

    /* zstdMode overrides gzipMode, as in conf.c:283-303 */
    s = ini_str(&ini, "PARAMETER:zstdMode", "Zstd_HIGH_SPEED");
    if (strcmp(s, "Zstd_BEST_SPEED") == 0) confparams_cpr->gzipMode = 1;
    else if (strcmp(s, "Zstd_HIGH_SPEED") == 0) confparams_cpr->gzipMode = 3;
    else if (strcmp(s, "Zstd_HIGH_COMPRESSION") == 0) confparams_cpr->gzipMode = 19;
    else if (strcmp(s, "Zstd_BEST_COMPRESSION") == 0) confparams_cpr->gzipMode = 22;
    else if (strcmp(s, "Zstd_DEFAULT_COMPRESSION") == 0) confparams_cpr->gzipMode = 3;
    else CONF_FAIL("[SZ] Error: Wrong zstd Mode (please check sz.config file)");

    s = ini_str(&ini, "PARAMETER:protectValueRange", "YES");
    confparams_cpr->protectValueRange = strcmp(s, "YES") == 0 ? 1 : 0;
    confparams_cpr->randomAccess = (int)ini_int(&ini, "PARAMETER:randomAccess", 0);
    confparams_cpr->snapshotCmprStep = (int)ini_int(&ini, "PARAMETER:snapshotCmprStep", 5);

    s = ini_str(&ini, "PARAMETER:errorBoundMode", NULL);
    if (!s) CONF_FAIL("[SZ] Error: Null error bound setting (please check sz.config file)");
    static const struct { const char *a, *b; int mode; } modes[] = {
        {"ABS", "abs", ABS}, {"REL", "rel", REL}, {"VR_REL", "vr_rel", REL}, {"ABS_AND_REL", "abs_and_rel", ABS_AND_REL},
        {"ABS_OR_REL", "abs_or_rel", ABS_OR_REL}, {"PW_REL", "pw_rel", PW_REL}, {"PSNR", "psnr", PSNR},
        {"ABS_AND_PW_REL", "abs_and_pw_rel", ABS_AND_PW_REL}, {"ABS_OR_PW_REL", "abs_or_pw_rel", ABS_OR_PW_REL},
        {"REL_AND_PW_REL", "rel_and_pw_rel", REL_AND_PW_REL}, {"REL_OR_PW_REL", "rel_or_pw_rel", REL_OR_PW_REL},
        {"NORM", "norm", NORM}};
    int found = 0;
    for (size_t i = 0; i < sizeof(modes) / sizeof(modes[0]); i++)
        if (strcmp(s, modes[i].a) == 0 || strcmp(s, modes[i].b) == 0) { confparams_cpr->errorBoundMode = modes[i].mode; found = 1; break; }
    if (!found) CONF_FAIL("[SZ] Error: Wrong error bound mode (please check sz.config file)");

    confparams_cpr->absErrBound = ini_dbl(&ini, "PARAMETER:absErrBound", 0);
    confparams_cpr->relBoundRatio = ini_dbl(&ini, "PARAMETER:relBoundRatio", 0);
    confparams_cpr->psnr = ini_dbl(&ini, "PARAMETER:psnr", 0);
    confparams_cpr->normErr = ini_dbl(&ini, "PARAMETER:normErr", 0);
    confparams_cpr->pw_relBoundRatio = ini_dbl(&ini, "PARAMETER:pw_relBoundRatio", 0);
    confparams_cpr->segment_size = (int)ini_int(&ini, "PARAMETER:segment_size", 0);
    confparams_cpr->accelerate_pw_rel_compression = (int)ini_int(&ini, "PARAMETER:accelerate_pw_rel_compression", 1);

    s = ini_str(&ini, "PARAMETER:pwr_type", "MIN");
    if (strcmp(s, "MIN") == 0) confparams_cpr->pwr_type = SZ_PWR_MIN_TYPE;
    else if (strcmp(s, "AVG") == 0) confparams_cpr->pwr_type = SZ_PWR_AVG_TYPE;
    else if (strcmp(s, "MAX") == 0) confparams_cpr->pwr_type = SZ_PWR_MAX_TYPE;
    else CONF_FAIL("[SZ] Error: Wrong pwr_type setting (please check sz.config file).");
#undef CONF_FAIL
    ini_free(&ini);
    return SZ_SCES;
}

int SZ_LoadConf(const char *sz_cfgFile)
{
    int res = SZ_ReadConf(sz_cfgFile);
    if (res != SZ_SCES) { printf("[SZ] ERROR: Impossible to read configuration.\n"); return SZ_NSCS; }
    return SZ_SCES;
}
