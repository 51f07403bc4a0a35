/*
 * plane_set.c -- per-aircraft state, the message validator / field decoder and
 * the printer behind reference src/ads-b/plane.h:12-38.  Host-side sequential
 * stage of the path: it sees only the frames the device accepted (or, in
 * SORA_ADSB_EMIT_ALL mode, every candidate) and re-applies the reference's
 * acceptance rule, so its verdict never depends on the device's.
 *
 * The message arrives as the reference's ASCII bit string (plane.h:16-17) and
 * is packed once into four MSB-first words; every field is then a shift and a
 * mask instead of a character loop (reference read_bit_string, plane.c:351-361).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sora_adsb_host.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum {
    HAVE_SQUAWK = 0x01, HAVE_CATEGORY = 0x02, HAVE_FLIGHT_ID = 0x04,
    HAVE_VELOCITY = 0x08, HAVE_ALTITUDE = 0x10, HAVE_GNSS_BARO = 0x20,
    HAVE_VERT_RATE = 0x40, HAVE_COORDS = 0x80	/* flag values of plane.c:21-28 */
};

struct plane {				/* fields of plane.c:17-53 */
    uint32_t address;
    int have;
    int category_set, category;
    int velocity_we, velocity_sn;	/* kt */
    long altitude;			/* ft */
    int gnss_vs_baro, vert_rate;
    unsigned cpr_lat[2], cpr_lon[2];	/* [0] even, [1] odd */
    struct timeval cpr_time[2];
    double latitude, longitude;
    struct timeval coordinates_time, last_update;
    char flight_id[9], squawk_id[5];
    double last_amplitude;
};

struct plane_set {
    struct plane **slot;		/* open addressing, power-of-two size */
    size_t size, used;
};

/* ---------------------------------------------------------- the set ---- */

struct plane_set *
plane_set_new(void) {
    struct plane_set *ps = calloc(1, sizeof *ps);

    if (ps == NULL)
	abort();		/* the reference raises memory_outage -> abort() */
    ps->size = 256;
    ps->slot = calloc(ps->size, sizeof *ps->slot);
    if (ps->slot == NULL)
	abort();
    return ps;
}

static void
set_release(struct plane_set *ps, int with_planes) {
    size_t i;

    if (ps == NULL)
	return;
    for (i = 0; with_planes && i < ps->size; i++)
	free(ps->slot[i]);
    free(ps->slot);
    free(ps);
}

void plane_set_delete(struct plane_set *ps) { set_release(ps, 1); }	/* plane.c:78-82 */
void plane_set_free(struct plane_set *ps) { set_release(ps, 0); }	/* plane.c:84-88 */

static size_t
probe_start(const struct plane_set *ps, uint32_t address) {
    return (address * 0x9e3779b1u >> 7) & (ps->size - 1);
}

static struct plane *
set_find(const struct plane_set *ps, uint32_t address) {
    size_t i = probe_start(ps, address);

    while (ps->slot[i] != NULL) {
	if (ps->slot[i]->address == address)
	    return ps->slot[i];
	i = (i + 1) & (ps->size - 1);
    }
    return NULL;
}

static void
set_insert(struct plane_set *ps, struct plane *pl) {
    size_t i;

    if ((ps->used + 1) * 2 > ps->size) {
	struct plane **old = ps->slot;
	size_t n = ps->size, k;

	ps->size *= 2;
	ps->slot = calloc(ps->size, sizeof *ps->slot);
	if (ps->slot == NULL)
	    abort();
	for (k = 0; k < n; k++)
	    if (old[k] != NULL) {
		i = probe_start(ps, old[k]->address);
		while (ps->slot[i] != NULL)
		    i = (i + 1) & (ps->size - 1);
		ps->slot[i] = old[k];
	    }
	free(old);
    }
    i = probe_start(ps, pl->address);
    while (ps->slot[i] != NULL)
	i = (i + 1) & (ps->size - 1);
    ps->slot[i] = pl;
    ps->used++;
}

static struct plane *
plane_make(uint32_t address) {
    struct plane *pl = calloc(1, sizeof *pl);	/* zeroed, plane.c:116 */

    if (pl == NULL)
	abort();
    pl->address = address;
    return pl;
}

struct plane *
plane_set_lookup_create(struct plane_set *ps, uint32_t address) {
    struct plane *pl = set_find(ps, address);

    if (pl == NULL) {
	pl = plane_make(address);
	set_insert(ps, pl);
    }
    return pl;
}

/* plane.h:18 over the open-addressing table: the occupied slots in table order */
struct set_iterator {
    struct iterator it;			/* first: the two pointers convert */
    struct plane_set *ps;
    size_t at;				/* next occupied slot, or ps->size */
};

static void
set_iterator_seek(struct set_iterator *si) {
    while (si->at < si->ps->size && si->ps->slot[si->at] == NULL)
	si->at++;
}

static int
set_iterator_is_at_end(struct iterator *it) {
    const struct set_iterator *si = (struct set_iterator *) it;

    return si->at >= si->ps->size;
}

static void
set_iterator_next_pair(struct iterator *it, const void **key, void **data) {
    struct set_iterator *si = (struct set_iterator *) it;

    if (si->at >= si->ps->size) {
	*key = *data = NULL;		/* hash.c:385-386 */
	return;
    }
    *key = (const void *) (intptr_t) si->ps->slot[si->at]->address;
    *data = si->ps->slot[si->at];
    si->at++;
    set_iterator_seek(si);
}

static void *
set_iterator_next(struct iterator *it) {
    const void *key;
    void *data;

    set_iterator_next_pair(it, &key, &data);
    return data;
}

static void
set_iterator_free(struct iterator *it) {
    free(it);
}

struct iterator *
plane_set_iterate(struct plane_set *ps) {
    struct set_iterator *si = calloc(1, sizeof *si);

    if (si == NULL)
	abort();
    si->it.is_at_end = set_iterator_is_at_end;
    si->it.next = set_iterator_next;
    si->it.next_pair = set_iterator_next_pair;
    si->it.free = set_iterator_free;
    si->ps = ps;
    set_iterator_seek(si);
    return &si->it;
}

struct plane *
plane_clone(struct plane *pl) {
    struct plane *copy = plane_make(pl->address);

    *copy = *pl;
    return copy;
}

void plane_delete(struct plane *pl) { free(pl); }

/* ------------------------------------------------------- the message ---- */

struct bits112 {
    uint32_t w[4];
};

static struct bits112
pack_message(const char *msg) {
    struct bits112 b = { { 0, 0, 0, 0 } };
    int k;

    /* like the reference, anything that is not '1' reads as 0, and a string
     * shorter than the field that is read is the caller's problem; here a
     * short string reads as zeros behind its end */
    for (k = 0; k < 112 && msg[k] != '\0'; k++)
	if (msg[k] == '1')
	    b.w[k >> 5] |= 1u << (31 - (k & 31));
    return b;
}

/* bits [off, off + n), n <= 32, MSB first */
static uint32_t
field(const struct bits112 *b, int off, int n) {
    uint64_t two = (uint64_t) b->w[off >> 5] << 32;

    if ((off >> 5) + 1 < 4)
	two |= b->w[(off >> 5) + 1];
    return (uint32_t) ((two << (off & 31)) >> (64 - n));
}

/* Mode S CRC-24 (generator 0x1FFF409, plane.c:373-385) over the first n bits */
static uint32_t
crc24(const struct bits112 *b, int n) {
    uint32_t r = 0;
    int k;

    for (k = 0; k < n; k++) {
	r = (r << 1) ^ (((b->w[k >> 5] >> (31 - (k & 31))) & 1u) << 24);
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r & 0xffffffu;
}

/* ----------------------------------------------------------------- CPR ---- */

static void
decode_position(struct plane *pl) {
    /* global airborne CPR from the latest even/odd pair, plane.c:387-450 */
    long long dt = (long long) pl->cpr_time[1].tv_sec - (long long) pl->cpr_time[0].tv_sec;
    int latest, j, zone, nl;
    double lat, lon;

    pl->have &= ~HAVE_COORDS;
    if (llabs(dt) > 10)
	return;
    latest = dt > 0;			/* 1: the odd half is the newer one */

    j = (int) floor((59 * (double) pl->cpr_lat[0] - 60 * (double) pl->cpr_lat[1]) /
	    (1 << 17) + 0.5);
    if (j < 0)
	j += 60 - latest;
    if (!((j >= 0 && j <= 15) || (j >= 45 && j <= 60)))
	return;
    if (j >= 45)
	j -= 60;
    lat = (j + (double) pl->cpr_lat[latest] / (1 << 17)) * 360 / (60 - latest);
    pl->latitude = lat;		/* stored even when the longitude step bails out */

    zone = (int) ((lat < 0? ceil(lat / 6) : floor(lat / 6)) * 6 + 3);
    zone = abs(zone);
    if (zone <= 3)
	nl = 59;
    else if (zone == 87)
	nl = 2;
    else if (zone > 87)
	nl = 1;
    else {
	double c = cos(M_PI * zone / 180);
	nl = (int) floor(2 * M_PI / acos(1 - (1 - cos(M_PI / 30)) / (c * c)));
    }

    j = (int) floor(((nl - 1) * (double) pl->cpr_lon[0] - nl * (double) pl->cpr_lon[1]) /
	    (1 << 17) + 0.5);
    if (j < 0)
	j += nl - latest;
    lon = (j + (double) pl->cpr_lon[latest] / (1 << 17)) * 360.0 / (nl - latest);
    if (lon >= 180)
	lon -= 360;
    pl->longitude = lon;
    pl->coordinates_time = pl->cpr_time[latest];
    pl->have |= HAVE_COORDS;
}

/* --------------------------------------------------------- the decoder ---- */

static char
callsign_character(uint32_t v) {		/* plane.c:337-349 */
    if (v >= 1 && v <= 26)
	return (char) ('A' + v - 1);
    if (v == 32)
	return ' ';
    if (v >= 48 && v <= 57)
	return (char) v;
    return '?';
}

#define BIT_OF(x, n)	(((x) >> (n)) & 1u)

/* three bits of a 13-bit AC/ID field, two apart (plane.c:366-371) */
static uint32_t
triplet(uint32_t x, int n, int msb_first) {
    uint32_t a = BIT_OF(x, n), b = BIT_OF(x, n + 2), c = BIT_OF(x, n + 4);

    return msb_first? (a << 2) | (b << 1) | c : a | (b << 1) | (c << 2);
}

static void
decode_df17(struct plane *pl, const struct bits112 *b, struct timeval tv) {
    uint32_t type = field(b, 32, 5);

    if (type >= 1 && type <= 4) {			/* plane.c:493-502 */
	int i;
	for (i = 0; i < 8; i++)
	    pl->flight_id[i] = callsign_character(field(b, 40 + 6 * i, 6));
	pl->flight_id[8] = '\0';
	pl->have |= HAVE_FLIGHT_ID;
    }
    if (type >= 9 && type <= 18) {			/* plane.c:503-519 */
	uint32_t alt = field(b, 40, 12);
	int odd = (int) field(b, 53, 1);

	if (BIT_OF(alt, 4)) {				/* Q bit: 25 ft steps */
	    pl->altitude = (long) ((((alt >> 5) & 0x7f) << 4) | (alt & 0xf)) * 25 - 1000;
	    pl->have |= HAVE_ALTITUDE;
	}
	pl->cpr_lat[odd] = field(b, 54, 17);
	pl->cpr_lon[odd] = field(b, 71, 17);
	pl->cpr_time[odd] = tv;
	decode_position(pl);
    } else if (type == 19) {				/* plane.c:522-550 */
	uint32_t subtype = field(b, 37, 3);

	if (subtype == 1 || subtype == 2) {
	    double we = field(b, 46, 10), sn = field(b, 57, 10);

	    if (we != 0 && sn != 0) {		/* 0 = no information */
		we -= 1;
		sn -= 1;
		if (field(b, 45, 1))
		    we = -we;
		if (field(b, 56, 1))
		    sn = -sn;
		if (subtype == 2) {		/* supersonic: 4 kt steps */
		    we *= 4;
		    sn *= 4;
		}
		pl->velocity_we = (int) we;
		pl->velocity_sn = (int) sn;
		pl->have |= HAVE_VELOCITY;
	    }
	}
    }
}

static void
decode_df4(struct plane *pl, uint32_t ac) {		/* plane.c:559-585 */
    static const int c_adjust[2][8] = {			/* plane.c:452-455 */
	{ -1, 0, 2, 1, 4, -1, 3, -1 },
	{ -1, 4, 2, 3, 0, -1, 1, -1 }
    };

    if (BIT_OF(ac, 6))				/* M = 1: metric, not decoded */
	return;
    if (BIT_OF(ac, 4)) {
	uint32_t n = (((ac >> 7) & 0x3f) << 5) | (BIT_OF(ac, 5) << 4) | (ac & 0xf);
	pl->altitude = (long) n * 25 - 1000;
	pl->have |= HAVE_ALTITUDE;
    } else {					/* Gillham code, 100 ft steps */
	uint16_t g = (uint16_t) (((triplet(ac, 0, 0) & 3) << 6) |
		(triplet(ac, 7, 0) << 3) | triplet(ac, 1, 0));
	int adjust;

	g ^= g >> 8;				/* Gray to binary, plane.c:327-335 */
	g ^= g >> 4;
	g ^= g >> 2;
	g ^= g >> 1;
	adjust = c_adjust[g & 1][triplet(ac, 8, 0)];
	if (adjust != -1) {
	    pl->altitude = (long) ((int) g * 5 - 12 + adjust) * 100;
	    pl->have |= HAVE_ALTITUDE;
	}
    }
}

struct plane *
plane_set_parse_message(struct plane_set *ps, const char *msg, struct timeval tv,
	double amplitude) {
    const struct bits112 b = pack_message(msg);
    const uint32_t df = field(&b, 0, 5);
    const int is_long = df >= 16 && df <= 24;		/* plane.c:462 */
    const uint32_t ap = field(&b, is_long? 88 : 32, 24);
    const uint32_t ac = field(&b, 19, 13);
    const uint32_t syndrome = crc24(&b, is_long? 88 : 32) ^ ap;
    struct plane *pl;

    if (df == 11 || df == 17 || df == 18) {		/* plane.c:473-478 */
	if (syndrome != 0)
	    return NULL;
	pl = plane_set_lookup_create(ps, field(&b, 8, 24));
    } else {						/* plane.c:479-484 */
	pl = set_find(ps, syndrome);
	if (pl == NULL)
	    return NULL;
    }

    pl->last_update = tv;
    pl->last_amplitude = amplitude;

    if (df == 17)
	decode_df17(pl, &b, tv);
    else if (df == 4)
	decode_df4(pl, ac);
    else if (df == 5) {					/* plane.c:586-593 */
	pl->squawk_id[0] = (char) ('0' + triplet(ac, 7, 1));
	pl->squawk_id[1] = (char) ('0' + triplet(ac, 1, 1));
	pl->squawk_id[2] = (char) ('0' + triplet(ac, 8, 1));
	pl->squawk_id[3] = (char) ('0' + triplet(ac, 0, 1));
	pl->squawk_id[4] = '\0';
	pl->have |= HAVE_SQUAWK;
    }
    /* DF 11, 18, 20, 21 and the rest only touch last_update / amplitude */
    return pl;
}

/* ------------------------------------------------------------ getters ---- */

const char *
plane_get_category(struct plane *pl) {			/* plane.c:137-235 */
    static const char *const names[3][8] = {
	{ "Cat A", "Light", "Medium 1", "Medium 2", "High vortex aircraft", "Heavy",
	  "High performance", "Rotorcraft" },
	{ "Cat B", "Glider/sailplane", "Lighter-than-air", "Parachutist/skydiver",
	  "Ultralight/hang-glider/paraglider", "Cat B5", "Unmanned Aerial Vehicle",
	  "Space/transatmospheric vehicle" },
	{ "Cat C", "Surface vehicle -- Emergency vehicle",
	  "Surface vehicle -- Service vehicle", "Fixed ground or tethered obstruction",
	  "Cat C4", "Cat C5", "Cat C6", "Cat C7" },
    };

    if (!(pl->have & HAVE_CATEGORY))
	return NULL;
    if (pl->category_set == 3)
	return "Cat D";
    if (pl->category_set < 0 || pl->category_set > 2 || pl->category < 0 ||
	    pl->category > 7)
	return NULL;
    return names[pl->category_set][pl->category];
}

const char *plane_get_squawk_id(struct plane *pl)
{ return (pl->have & HAVE_SQUAWK)? pl->squawk_id : NULL; }
const char *plane_get_flight_id(struct plane *pl)
{ return (pl->have & HAVE_FLIGHT_ID)? pl->flight_id : NULL; }
int plane_is_ground_velocity_available(struct plane *pl)
{ return !!(pl->have & HAVE_VELOCITY); }
int plane_is_heading_available(struct plane *pl)
{ return !!(pl->have & HAVE_VELOCITY); }
double plane_get_ground_velocity(struct plane *pl) { (void) pl; return 0; }	/* plane.c:263-267 */
double plane_get_heading(struct plane *pl) { (void) pl; return 0; }
int plane_is_altitude_available(struct plane *pl) { return !!(pl->have & HAVE_ALTITUDE); }
long plane_get_altitude(struct plane *pl) { return pl->altitude; }
int plane_is_gnss_vs_baro_available(struct plane *pl) { return !!(pl->have & HAVE_GNSS_BARO); }
int plane_get_gnss_vs_baro(struct plane *pl) { return pl->gnss_vs_baro; }
int plane_is_vert_rate_available(struct plane *pl) { return !!(pl->have & HAVE_VERT_RATE); }
int plane_get_vert_rate(struct plane *pl) { return pl->vert_rate; }
int
plane_is_lat_long_available(struct plane *pl) {		/* plane.c:305-311 */
    return pl->cpr_time[0].tv_sec != 0 && pl->cpr_time[0].tv_usec != 0 &&
	    pl->cpr_time[1].tv_sec != 0 && pl->cpr_time[1].tv_usec != 0;
}
double plane_get_latitude(struct plane *pl) { (void) pl; return 0; }	/* plane.c:313-323 */
double plane_get_longitude(struct plane *pl) { (void) pl; return 0; }

/* ------------------------------------------------------------ printer ---- */

int
plane_snprint(struct plane *pl, char *buf, size_t len) {
    /* the fixed-width line of plane.c:598-632 */
    size_t o = 0;
    struct tm *tm;
    time_t t;

#define PUT(...) do { \
	int n_ = snprintf(buf + (o < len? o : len), o < len? len - o : 0, __VA_ARGS__); \
	if (n_ > 0) o += (size_t) n_; \
    } while (0)
    PUT("%06X ", (unsigned) pl->address);
    if (pl->have & HAVE_FLIGHT_ID) PUT("%-8s ", pl->flight_id); else PUT("%9s", "");
    if (pl->have & HAVE_SQUAWK) PUT("%4s ", pl->squawk_id); else PUT("%5s", "");
    if (pl->have & HAVE_ALTITUDE) PUT("%5ldft ", pl->altitude); else PUT("%8s", "");
    if (pl->have & HAVE_VELOCITY) {
	double we = pl->velocity_we, sn = pl->velocity_sn;
	double heading = atan2(we, sn) * 180 / M_PI;
	if (heading < 0)
	    heading += 360;
	PUT("%6.1fkt %05.1fd ", sqrt(we * we + sn * sn), heading);
    } else
	PUT("%16s", "");
    if (pl->have & HAVE_COORDS) PUT("% 8.5f % 8.5f ", pl->latitude, pl->longitude);
    else PUT("%19s", "");
    t = pl->last_update.tv_sec;
    tm = localtime(&t);
    PUT("%02d:%02d:%02d", tm->tm_hour, tm->tm_min, tm->tm_sec);
    PUT(" %5.1f", pl->last_amplitude);
#undef PUT
    return (int) o;
}

void
plane_print(struct plane *pl) {
    char line[256];

    plane_snprint(pl, line, sizeof line);
    fputs(line, stdout);
}
