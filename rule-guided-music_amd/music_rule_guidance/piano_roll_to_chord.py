"""Piano roll -> note events -> Standard MIDI File, without pretty_midi / mido (neither is vendored).

Reference: music_rule_guidance/piano_roll_to_chord.py:167-275 (piano_roll_to_pretty_midi: which notes and sustain-pedal
events a generated (3,128,T) roll [velocity | onset | pedal] turns into) and guided_diffusion/midi_util.py:67-93
(save_piano_roll_midi).  The event extraction is restated here and pinned to the reference's output
(tests/golden/midi_events.npz).  The container classes mirror the pretty_midi attributes the reference touches
(`.instruments[0].notes`, `.control_changes`, `.write(path)`).  The writer's whole MESSAGE STREAM -- tick conversion, the order of
events at equal ticks, channels, track layout, delta ticks -- is pinned to what the reference's vendored pretty_midi fork hands to
mido (tests/golden/midi_writer.npz, tests/test_host_logic.py); only mido's byte serialisation of those messages is this module's own,
following the SMF 1.0 specification (format 1, 220 ticks per quarter note at 120 bpm -- pretty_midi's defaults).  Chord analysis
(music21) stays a host plug-in: music_rules.register_chord_backend (blocking get_chords, or get_chords_async beside the GPU in the
samplers' search step).  Host-side I/O only -- nothing here is on the sampling hot path.
"""
import math
import struct

import numpy as np

from .music_rules import MAX_PIANO, MIN_PIANO

RESOLUTION = 220            # ticks per quarter note
TEMPO_US = 500000           # 120 bpm
TICKS_PER_SECOND = RESOLUTION * 1e6 / TEMPO_US


class Note:
    __slots__ = ("velocity", "pitch", "start", "end")

    def __init__(self, velocity, pitch, start, end):
        self.velocity, self.pitch, self.start, self.end = int(velocity), int(pitch), float(start), float(end)

    def __repr__(self):
        return f"Note(start={self.start:f}, end={self.end:f}, pitch={self.pitch}, velocity={self.velocity})"


class ControlChange:
    __slots__ = ("number", "value", "time")

    def __init__(self, number, value, time):
        self.number, self.value, self.time = int(number), int(value), float(time)


class Instrument:
    def __init__(self, program=0, is_drum=False, name=""):
        self.program, self.is_drum, self.name = int(program), bool(is_drum), name
        self.notes, self.control_changes = [], []


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track(events):
    """events: (tick, order, bytes) -> MTrk chunk with delta times and the end-of-track meta event."""
    body, last = bytearray(), 0
    for tick, _, data in sorted(events, key=lambda e: (e[0], e[1])):
        body += _vlq(tick - last) + data
        last = tick
    body += _vlq(1 if events else 0) + b"\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + bytes(body)


class SimpleMIDI:
    """The slice of pretty_midi.PrettyMIDI the sampling scripts use: a list of instruments, write(), and a reader."""

    def __init__(self, midi_file=None):
        self.instruments = []
        self.resolution = RESOLUTION
        if midi_file is not None:
            self._read(midi_file)

    def get_end_time(self):
        ends = [n.end for i in self.instruments for n in i.notes] + [c.time for i in self.instruments for c in i.control_changes]
        return max(ends) if ends else 0.0

    # ------------------------------------------------------------------ writer
    # Message stream = what the reference's vendored pretty_midi fork hands to mido (pretty_midi/pretty_midi.py:1341-1520), pinned by
    # tests/golden/midi_writer.npz (a recording stand-in for mido around the fork's own write()): tick conversion, event order, channels,
    # track layout, delta ticks.  Only mido's byte serialisation (the SMF standard: _vlq and write() below) is this module's own.
    MSG_TIME_SIGNATURE, MSG_SET_TEMPO, MSG_PROGRAM, MSG_CONTROL, MSG_NOTE_ON, MSG_END = 0, 1, 2, 3, 4, 5

    def time_to_tick(self, t):
        """pretty_midi.time_to_tick (:1077-1109) of a freshly built PrettyMIDI: one tempo, so tick = round(t / seconds per tick) with
        Python's round (halves to even), the division written as the fork writes it."""
        if t <= 0.0:
            return 0
        return int(round(t / (60.0 / (120.0 * self.resolution))))

    def messages(self):
        """[(track, type, channel, data1, data2, delta_ticks)] in file order (types: MSG_*).  Track 0 = tempo + 4/4; one track per
        instrument: program change, then all events sorted by (tick, class, data) as the fork's comparator does (:1350-1394) -- at equal
        ticks control changes by (control, value), then note-ons by (pitch, velocity), a note-off being a note-on of velocity 0 -- and an
        end-of-track one tick behind the last event.  One deviation, outside what the samplers produce (their notes last >= 1 column =
        4.4 ticks): a note shorter than a tick ends one tick after it starts (the fork would emit its off BEFORE its on: a stuck note)."""
        tempo = int(6e7 / (60. / ((60.0 / (120.0 * self.resolution)) * self.resolution)))
        out = [(0, self.MSG_SET_TEMPO, 0, tempo, 0, 0), (0, self.MSG_TIME_SIGNATURE, 0, 4, 4, 0), (0, self.MSG_END, 0, 0, 0, 1)]
        for k, ins in enumerate(self.instruments):
            ch = 9 if ins.is_drum else (k % 15 if k % 15 < 9 else k % 15 + 1)
            ev = [(0, 6 << 16, self.MSG_PROGRAM, ins.program & 0x7F, 0)]
            for n in ins.notes:
                t0 = self.time_to_tick(n.start)
                t1 = max(self.time_to_tick(n.end), t0 + 1)
                vel = max(1, n.velocity) & 0x7F
                ev.append((t0, (10 << 16) + ((n.pitch & 0x7F) << 8) + vel, self.MSG_NOTE_ON, n.pitch & 0x7F, vel))
                ev.append((t1, (10 << 16) + ((n.pitch & 0x7F) << 8), self.MSG_NOTE_ON, n.pitch & 0x7F, 0))
            for c in ins.control_changes:
                ev.append((self.time_to_tick(c.time), (8 << 16) + ((c.number & 0x7F) << 8) + (c.value & 0x7F), self.MSG_CONTROL, c.number & 0x7F, c.value & 0x7F))
            ev.sort(key=lambda e: (e[0], e[1]))              # stable, like sorted(cmp_to_key(event_compare))
            last = 0
            for tick, _, kind, a, b in ev:
                out.append((k + 1, kind, ch, a, b, tick - last))
                last = tick
            out.append((k + 1, self.MSG_END, 0, 0, 0, 1))
        return out

    def write(self, filename):
        tracks = {}
        for trk, kind, ch, a, b, delta in self.messages():
            body = tracks.setdefault(trk, bytearray())
            body += _vlq(delta)
            if kind == self.MSG_SET_TEMPO:
                body += b"\xff\x51\x03" + struct.pack(">I", a)[1:]
            elif kind == self.MSG_TIME_SIGNATURE:
                body += bytes([0xFF, 0x58, 0x04, a, {1: 0, 2: 1, 4: 2, 8: 3, 16: 4}[b], 24, 8])
            elif kind == self.MSG_PROGRAM:
                body += bytes([0xC0 | ch, a])
            elif kind == self.MSG_CONTROL:
                body += bytes([0xB0 | ch, a, b])
            elif kind == self.MSG_NOTE_ON:
                body += bytes([0x90 | ch, a, b])
            else:
                body += b"\xff\x2f\x00"
        with open(filename, "wb") as f:
            f.write(b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), self.resolution))
            for trk in sorted(tracks):
                f.write(b"MTrk" + struct.pack(">I", len(tracks[trk])) + bytes(tracks[trk]))

    # ------------------------------------------------------------------ reader (format 0 / 1, tempo map, running status)
    def _read(self, filename):
        with open(filename, "rb") as f:
            data = f.read()
        if data[:4] != b"MThd":
            raise ValueError(f"{filename}: not a Standard MIDI File")
        hlen, _fmt, ntrk, div = struct.unpack(">IHHH", data[4:14])
        if div & 0x8000:
            raise NotImplementedError("SMPTE time division")
        self.resolution = div
        pos = 8 + hlen
        tracks, tempos = [], [(0, TEMPO_US)]
        for _ in range(ntrk):
            if data[pos:pos + 4] != b"MTrk":
                raise ValueError(f"{filename}: bad track chunk")
            tlen = struct.unpack(">I", data[pos + 4:pos + 8])[0]
            tr, p, end, tick, status = [], pos + 8, pos + 8 + tlen, 0, 0
            while p < end:
                d = 0
                while True:
                    b = data[p]
                    p += 1
                    d = (d << 7) | (b & 0x7F)
                    if not b & 0x80:
                        break
                tick += d
                b = data[p]
                if b == 0xFF:
                    kind = data[p + 1]
                    p += 2
                    ln = 0
                    while True:
                        c = data[p]
                        p += 1
                        ln = (ln << 7) | (c & 0x7F)
                        if not c & 0x80:
                            break
                    if kind == 0x51 and ln == 3:
                        tempos.append((tick, int.from_bytes(data[p:p + 3], "big")))
                    p += ln
                elif b in (0xF0, 0xF7):
                    p += 1
                    ln = 0
                    while True:
                        c = data[p]
                        p += 1
                        ln = (ln << 7) | (c & 0x7F)
                        if not c & 0x80:
                            break
                    p += ln
                else:
                    if b & 0x80:
                        status = b
                        p += 1
                    hi = status & 0xF0
                    nargs = 1 if hi in (0xC0, 0xD0) else 2
                    a = data[p:p + nargs]
                    p += nargs
                    tr.append((tick, status, a[0], a[1] if nargs == 2 else 0))
            tracks.append(tr)
            pos = end
        tempos.sort()
        t_ticks = np.array([t for t, _ in tempos], dtype=np.float64)
        t_us = np.array([u for _, u in tempos], dtype=np.float64)
        t_sec = np.concatenate(([0.0], np.cumsum(np.diff(t_ticks) * t_us[:-1] / (1e6 * div))))

        def seconds(tick):
            i = int(np.searchsorted(t_ticks, tick, side="right") - 1)
            return float(t_sec[i] + (tick - t_ticks[i]) * t_us[i] / (1e6 * div))

        for tr in tracks:
            by_ch = {}
            for tick, status, a, b in tr:
                ch, hi = status & 0x0F, status & 0xF0
                ins, on = by_ch.setdefault(ch, (Instrument(0, ch == 9), {}))
                if hi == 0xC0:
                    ins.program = a
                elif hi == 0xB0:
                    ins.control_changes.append(ControlChange(a, b, seconds(tick)))
                elif hi == 0x90 and b > 0:
                    on.setdefault(a, []).append((tick, b))
                elif hi == 0x80 or (hi == 0x90 and b == 0):
                    if on.get(a):
                        t0, vel = on[a].pop(0)
                        ins.notes.append(Note(vel, a, seconds(t0), seconds(tick)))
            for ch in sorted(by_ch):
                ins = by_ch[ch][0]
                if ins.notes or ins.control_changes:
                    ins.notes.sort(key=lambda n: (n.start, n.pitch))
                    self.instruments.append(ins)


def piano_roll_to_pretty_midi(full_roll, fs=100, program=0):
    """(128,T), (2,128,T) [velocity | pedal] or (3,128,T) [velocity | onset | pedal] roll in [0,127] -> SimpleMIDI with
    one instrument (reference :167-275).  Like the reference this WRITES into full_roll: onsets < 64 and pedal < 4
    are zeroed, velocities <= the loudest value below the piano range become 0.

    A note spans a maximal run of non-zero velocity in a pitch row and takes the run's first velocity; with an onset
    channel the run is cut at every onset inside it (+1 column) and dropped when it has none.  Pedal: the piano rows'
    mean per column (truncated), emitted as CC 64 where non-zero, < 16 -> 0 and > 112 -> 127."""
    full_roll = np.asarray(full_roll)
    onset_roll = None
    if full_roll.ndim == 3:
        piano_roll = full_roll[0]
        if full_roll.shape[0] == 2:
            pedal_roll = full_roll[1]
        else:
            onset_roll = full_roll[1]
            onset_roll[onset_roll < 64] = 0
            pedal_roll = full_roll[2]
        pedal_roll[pedal_roll < 4] = 0
        pedal = pedal_roll[MIN_PIANO:MAX_PIANO + 1].mean(axis=0).astype(np.intc)
        has_pedal = not math.isclose(pedal.max(), 0)
    else:
        piano_roll, pedal, has_pedal = full_roll, None, False
    piano_roll[piano_roll <= piano_roll[:MIN_PIANO, :].max()] = 0
    ins = Instrument(program=program)
    T = piano_roll.shape[1]
    active = np.zeros((128, T + 2), dtype=np.int8)
    active[:, 1:-1] = piano_roll != 0
    edges = np.diff(active, axis=1)                        # +1 at the first column of a run, -1 one past its last column
    off_t, off_p = np.nonzero(edges.T == -1)               # note-offs in (time, pitch) order: the order the reference appends in
    starts = {}
    on_t, on_p = np.nonzero(edges.T == 1)
    for t, p in zip(on_t, on_p):
        starts.setdefault(int(p), []).append(int(t))
    taken = {p: 0 for p in starts}
    for t, p in zip(off_t, off_p):
        t, p = int(t), int(p)
        s = starts[p][taken[p]]
        taken[p] += 1
        vel = int(piano_roll[p, s])
        if onset_roll is None:
            ins.notes.append(Note(vel, p, s / fs, t / fs))
            continue
        s_ind, e_ind = round((s / fs) * fs), round((t / fs) * fs)
        ons = np.nonzero(onset_roll[p, s_ind:e_ind + 1])[0]
        if len(ons) == 0:
            continue
        st = (ons + s_ind) / fs
        en = np.concatenate((st[1:], np.array([t / fs])))
        for a, b in zip(st, en):
            ins.notes.append(Note(vel, p, a, b))
    if has_pedal:
        for t in np.nonzero(pedal)[0]:
            v = int(pedal[t])
            v = 0 if v < 16 else (127 if v > 112 else v)
            ins.control_changes.append(ControlChange(64, v, t / fs))
    pm = SimpleMIDI()
    pm.instruments.append(ins)
    return pm


def quantize_pedal(value, num_bins=8):
    """pedal CC value 0..127 -> centre of its bin (reference midi_util.py:252-264: 0..15 -> 8, ..., 112..127 -> 120)."""
    if value < 0 or value > 127:
        raise ValueError("Value should be between 0 and 127")
    width = 128 // num_bins
    return min(int(value) // width * width + width // 2, 127)


def midi_to_full_piano_roll(pm, fs=100):
    """SimpleMIDI (or anything with .instruments of notes / control_changes) -> (3,128,T) [velocity | onset | pedal] float32 roll:
    the reference's get_full_piano_roll (midi_util.py:267-291) over its vendored pretty_midi FORK's
    get_piano_roll(fs, pedal_threshold=None, onset=True) (/root/reference/pretty_midi/instrument.py:70-205, pretty_midi.py:797-852),
    restated and pinned to both (tests/golden/midi_rolls.npz):
      * per instrument: no notes -> nothing (the fork's onset=True path cannot unpack such an instrument and raises); int(fs * end_time) columns, end_time = last note end / control change of THAT instrument;
        drums -> zeros; velocity += over [int(start*fs), int(end*fs)) (overlaps add up, nothing is clipped; a note shorter than a
        column leaves no velocity); onset = 127 (a binary mark, not the velocity) at min(int(start*fs), columns - 1);
      * instruments are summed into max-columns rolls, the onset roll clipped to 127;
      * pedal: the reference's own loop (quantised CC 64 values on the piano rows, a 0 -> 127 jump in one column shifted by two)."""
    rolls = []
    for ins in pm.instruments:
        if not ins.notes:
            continue
        ends = [n.end for n in ins.notes] + [c.time for c in ins.control_changes] + [b.time for b in getattr(ins, "pitch_bends", [])]
        cols = int(fs * max(ends))
        vel, on = np.zeros((128, cols)), np.zeros((128, cols))
        if not ins.is_drum:
            for n in ins.notes:
                vel[n.pitch, int(n.start * fs):int(n.end * fs)] += n.velocity
                if cols > 0:
                    on[n.pitch, min(int(n.start * fs), cols - 1)] = 127
        rolls.append((vel, on))
    T = max([v.shape[1] for v, _ in rolls], default=0)
    if T == 0:   # no instrument has a note that reaches the first column: the reference's fork raises here; a (3,128,0) roll would fail far from its cause
        raise ValueError("midi_to_full_piano_roll: no notes (or none lasting one column of 1/fs s) -- the reference cannot build a roll from this file either")
    roll = np.zeros((3, 128, T), dtype=np.float64)
    for v, o in rolls:
        roll[0, :, :v.shape[1]] += v
        roll[1, :, :o.shape[1]] += o
    np.clip(roll[1], 0, 127, out=roll[1])
    for ins in pm.instruments:
        for cc in ins.control_changes:
            if cc.number != 64:
                continue
            t = int(cc.time * fs)
            if t < T:
                if roll[2, MIN_PIANO, t] != 0.0 and abs(roll[2, MIN_PIANO, t] - cc.value) > 64:
                    roll[2, MIN_PIANO:MAX_PIANO + 1, min(t + 2, T - 1)] = quantize_pedal(cc.value)
                else:
                    roll[2, MIN_PIANO:MAX_PIANO + 1, t] = quantize_pedal(cc.value)
    return roll.astype(np.float32)
