/* ORACLE -- TEST INFRASTRUCTURE ONLY (see aerol_oracle.c). */
#ifndef AEROL_ORACLE_H
#define AEROL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct jo_aerol jo_aerol;
/* = AeroL(parent) + setSettings(fb, burstmode=false); fb in {600, 1200, 10500} (P channel) or 8400 (C channel, DecodeC:
 * jo_aerol_take_sus rows are the three sub-band signal units of a frame [frame, k, 12 bytes, crc_ok, 0]) */
jo_aerol *jo_aerol_create(int fb);
jo_aerol *jo_aerol_create_burst(int fb); /* setSettings(fb, burstmode = true): R/T packets, 10500 bps only */
long jo_aerol_take_packets(jo_aerol *a, int32_t *dst, long caprows); /* rows of 16 int32: packet, chunk, 12 bytes, total bytes, type (1 R, 2 T) */
void jo_aerol_destroy(jo_aerol *a);
/* = processDemodulatedSoftBits(soft_bits): Decode(bits, soft=true), continuous (P-channel) path */
void jo_aerol_write(jo_aerol *a, const int16_t *soft, long n);
/* signal units as the frame loop produces them (aerol.cpp:1583-1600): rows of 16 int32
 * [frame number (count of completed frames), k, byte0..byte11, crc_ok, frameinfo used for this frame] */
long jo_aerol_take_sus(jo_aerol *a, int32_t *dst, long caprows);
/* rows of 3 int64: [index of the soft bit, kind, value]; kind 0 = DataCarrierDetect(value), 1 = "Error short frame" , 2 = gotsync */
long jo_aerol_take_events(jo_aerol *a, int64_t *dst, long caprows);
/* C channel: one row of 304 bytes per frame: uint32 frame number, then the 300 voice bytes handed to Voicesignal (aerol.cpp:2454-2481) */
long jo_aerol_take_voice(jo_aerol *a, unsigned char *dst, long caprows);
int jo_aerol_dcd(jo_aerol *a);
int jo_aerol_tick_dcd(jo_aerol *a); /* AeroL::updateDCD, returns the data-carrier-detect flag afterwards */
#ifdef __cplusplus
}
#endif
#endif
