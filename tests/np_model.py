"""An independent numpy-float32 reading of the grail-rs iterator chain, written as Python
generators straight from reference src/lib.rs (Selector :990, Sequencer :859, Jitter :753,
Synthesize :497, ValueNoise :227, ArrayValueNoise :270).  It shares no code with the C
oracle (oracle/grail_oracle.c); tests cross-check the two bitwise on short tracks.
TEST INFRASTRUCTURE ONLY.  Slow (Python loop per sample): keep tracks to a few thousand samples.
"""
import numpy as np

f32 = np.float32
NF = 8
ONE, HALF, TWO = f32(1.0), f32(0.5), f32(2.0)


def random_f32(state):
    """:36-55 -> (value, new_state)"""
    state = (state * 16807 + 1) & 0xFFFFFFFF
    bits = np.uint32((state >> 9) | 0x3F800000)
    return (bits.view(f32) - f32(1.5)) * TWO, state


def tan_approx(x):  # :63-70, x: f32 array
    return ((ONE - x) * x * (f32(5.0) - f32(4.0) * (x + HALF) * (HALF - x))) / (
        (x + HALF) * (f32(5.0) - f32(4.0) * (ONE - x) * x) * (HALF - x))


def exp_approx(x):  # :75-82
    o = ONE - x
    o2 = o * o
    return o2 * o2 * o


def array_sum(a):  # :123-125 sequential fold from 0.0
    s = f32(0.0)
    for v in a:
        s = f32(s + v)
    return s


class Elem:
    """SynthesisElem :316-337"""
    __slots__ = ("frequency", "freq", "bw", "smooth", "breath", "turb", "amp")

    def __init__(self, frequency, freq, bw, smooth, breath, turb, amp):
        self.frequency = f32(frequency)
        self.freq, self.bw, self.smooth = (np.array(a, dtype=f32) for a in (freq, bw, smooth))
        self.breath, self.turb, self.amp = (np.array(a, dtype=f32) for a in (breath, turb, amp))

    @classmethod
    def from_floats(cls, a):
        a = np.asarray(a, dtype=f32)
        return cls(a[0], a[1:9], a[9:17], a[17:25], a[25:33], a[33:41], a[41:49])

    def floats(self):
        return np.concatenate([[self.frequency], self.freq, self.bw, self.smooth, self.breath,
                               self.turb, self.amp]).astype(f32)

    @classmethod
    def silent(cls):  # :367-377
        q, z = [0.25] * NF, [0.0] * NF
        return cls(0.25, q, q, q, z, z, z)

    def copy_silent(self):  # :454-459
        return Elem(self.frequency, self.freq, self.bw, self.smooth, self.breath, self.turb,
                    np.zeros(NF, dtype=f32))

    def copy_with_frequency(self, frequency):  # :445-450
        fr = f32(frequency)
        fr = f32(0.5) if (np.isnan(fr) or fr > f32(0.5)) else fr  # f32::min: NaN loses
        return Elem(fr, self.freq, self.bw, self.smooth, self.breath, self.turb, self.amp)

    def blend(self, other, alpha):  # :404-414: self*(1-alpha) + other*alpha
        a = f32(alpha)
        oma = f32(ONE - a)

        def b(x, y):
            return x * oma + y * a
        return Elem(b(self.frequency, other.frequency), b(self.freq, other.freq),
                    b(self.bw, other.bw), b(self.smooth, other.smooth),
                    b(self.breath, other.breath), b(self.turb, other.turb),
                    b(self.amp, other.amp))


def f32_min(a, b):
    """Rust f32::min: if one is NaN return the other."""
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a < b else b


def selector(phoneme_elems, voice_phonemes):
    """:990-1005. phoneme_elems: (phoneme, length, blend_length, frequency); voice_phonemes:
    dict phoneme -> Elem (VoiceStorage::get: missing = None)."""
    for ph, length, blend_length, frequency in phoneme_elems:
        e = voice_phonemes.get(int(ph))
        yield (e.copy_with_frequency(frequency) if e is not None else None, f32(length),
               f32(blend_length))


def sequencer(seq_elems, sample_rate):
    """:859-932. seq_elems yields (Elem|None, length, blend_length)."""
    it = iter(seq_elems)
    dt = f32(ONE / f32(sample_rate))  # :944
    cur = nxt = None
    time = f32(0.0)
    with np.errstate(all="ignore"):
        while True:
            time = f32(time - dt)  # :861
            if time < 0:  # :864
                if cur is not None and nxt is not None:  # :868
                    a = nxt
                    cur = nxt
                    nxt = next(it, None)
                    time = f32(time + a[1])
                elif cur is None and nxt is None:  # :876
                    cur = next(it, None)
                    nxt = next(it, None)
                    if cur is not None:
                        time = f32(time + cur[1])
                else:
                    return  # :886
            if cur is None:
                return  # :930
            b = cur[0]
            c = nxt[0] if nxt is not None else None
            if b is not None and c is not None:  # :897-903
                alpha = f32_min(f32(time / cur[2]), ONE)
                yield c.blend(b, alpha)
            elif b is not None:  # :906-912
                alpha = f32_min(f32(time / cur[2]), ONE)
                yield b.copy_silent().blend(b, alpha)
            elif c is not None:  # :915-921
                alpha = f32_min(f32(time / cur[2]), ONE)
                yield c.blend(c.copy_silent(), alpha)
            else:  # :924-927
                yield Elem.silent()


class ValueNoise:  # :218-255
    def __init__(self, state):
        self.current, state = random_f32(state)
        self.next, state = random_f32(state)
        self.phase = f32(0.0)
        self.state = state
        self.out_state = state

    def step(self, increment):
        self.phase = f32(self.phase + increment)
        if self.phase > ONE:
            self.phase = f32(self.phase - ONE)
            self.current = self.next
            self.next, self.state = random_f32(self.state)
        return f32(f32(self.current * f32(ONE - self.phase)) + f32(self.next * self.phase))


class ArrayValueNoise:  # :261-306
    def __init__(self, state):
        cur, nxt = np.zeros(NF, dtype=f32), np.zeros(NF, dtype=f32)
        for i in range(NF):  # :275-278 interleaved draws
            cur[i], state = random_f32(state)
            nxt[i], state = random_f32(state)
        self.current, self.next, self.phase, self.state = cur, nxt, f32(0.0), state
        self.out_state = state

    def step(self, increment):
        self.phase = f32(self.phase + increment)
        if self.phase > ONE:
            self.phase = f32(self.phase - ONE)
            self.current = self.next
            new = np.zeros(NF, dtype=f32)
            for i in range(NF):  # from_func order :301
                new[i], self.state = random_f32(self.state)
            self.next = new
        return self.current * f32(ONE - self.phase) + self.next * self.phase  # :305


def jitter(elems, seed, voice_scalars):
    """:753-797. voice_scalars: (jitter_frequency, d_frequency, d_formant_frequency, d_amplitude)."""
    jf, dfreq, dff, damp = (f32(x) for x in voice_scalars)
    freq_noise = ValueNoise(int(seed) & 0xFFFFFFFF)  # :789
    formant_freq_noise = ArrayValueNoise(freq_noise.out_state)  # :790 (same &mut seed)
    formant_amp_noise = ArrayValueNoise(formant_freq_noise.out_state)  # :791
    for e in elems:
        fr = freq_noise.step(jf)
        ffr = formant_freq_noise.step(jf)
        fam = formant_amp_noise.step(jf)
        frequency = f32(e.frequency + f32(fr * dfreq))  # :763
        freq = e.freq + ffr * dff  # :764
        delta = (fam + ONE) * f32(HALF * damp)  # :768-769
        mul = ONE - delta  # :772
        yield Elem(frequency, freq, e.bw, e.smooth, e.breath, e.turb, e.amp * mul)


def synthesize(elems):
    """:497-578"""
    phase = f32(0.0)
    a = np.zeros(NF, dtype=f32)
    b = np.zeros(NF, dtype=f32)
    c = np.zeros(NF, dtype=f32)
    seed = 0
    with np.errstate(all="ignore"):
        for e in elems:
            f = e.frequency
            if phase < f:  # :503-514
                t = f32(phase / f)
                polyblep = f32(f32(f32(TWO * t) - f32(t * t)) - ONE)
            elif phase > f32(ONE - f):
                t = f32(f32(phase - ONE) / f)
                polyblep = f32(f32(f32(t * t) + f32(TWO * t)) + ONE)
            else:
                polyblep = f32(0.0)
            saw = f32(f32(f32(TWO * phase) - ONE) - polyblep)  # :517
            phase = f32(phase + f)  # :520
            if phase >= ONE:
                phase = f32(phase - ONE)
            noise, seed = random_f32(seed)  # :528
            noise_wave = saw * (ONE - e.breath) + noise * e.breath  # :531
            alpha = exp_approx(e.smooth)  # :535
            a = a + (ONE - alpha) * (noise_wave - a)  # :538
            turbulence = a * (ONE * (ONE - e.turb) + noise * e.turb)  # :544-545
            v0 = turbulence * e.amp  # :550
            g = tan_approx(e.freq)  # :555
            k = e.bw / e.freq  # :558
            a1 = ONE / (ONE + g * (g + k))  # :560
            a2 = g * a1
            a3 = g * a2
            v3 = v0 - c  # :565
            v1 = a1 * b + a2 * v3
            v2 = c + a2 * b + a3 * v3
            b = TWO * v1 - b  # :570
            c = TWO * v2 - c
            yield f32(array_sum(v1) * HALF)  # :574


def voice_tables(voice):
    """voice: a ctypes Voice (grail_hip or oracle_lib layout) -> (phoneme dict, scalars, rate)."""
    raw = np.frombuffer(bytes(voice), dtype=f32)
    ph = {3: Elem.from_floats(raw[1:50]), 4: Elem.from_floats(raw[50:99])}
    return ph, tuple(raw[100:104]), raw[0]


def render(voice, phoneme_elems, seed=0):
    """phoneme_elems.select(v).sequence(v).jitter(seed, v).synthesize().collect()"""
    ph, scalars, rate = voice_tables(voice)
    chain = synthesize(jitter(sequencer(selector(phoneme_elems, ph), rate), seed, scalars))
    return np.array(list(chain), dtype=f32)


def trace(voice, phoneme_elems, seed=0, stage=1):
    ph, scalars, rate = voice_tables(voice)
    seq = sequencer(selector(phoneme_elems, ph), rate)
    src = seq if stage == 0 else jitter(seq, seed, scalars)
    return np.array([e.floats() for e in src], dtype=f32)
