"""CPU model of the history hand-over in k_msk_fb with a three-way split (jaero_amd/csrc/k_msk_fb.h: MFB4_* for the 80-tap filter with four
pairs per workgroup, MFB2_* for the 160-tap filter with two).

The 80-entry matched-filter history of an arm is split three ways: the 32 newest entries in LDS and the next 26 in the front half's
registers, the 18 oldest (22 until round 5) in the back half's registers (160 entries: 72 / 36 / 52).  The back half starts every filter sum (oldest first) two samples ahead and hands
it over; the front half announces, two samples ahead, the entry that will reach the back half's tail.  Across launches the pending sum
travels in the state (S_MFB_A0_*), the tails in firsave.  The kernel runs on the GPU only (tests/test_gpu_parity.py, test_gpu_scale.py);
this model replays its schedule -- mailbox slots, barriers as phase boundaries, launches of 0, 1, 2 ... samples -- with integer data
(so that every order of summation gives the same number) and checks each output against the plain 80-tap sum."""
import numpy as np
import pytest

FIRN, L, TB = 80, 32, 18
TF = FIRN - L - TB


class Pair:
    def __init__(self, taps):
        self.taps = taps
        self.lds = [0] * L          # ring, slot fir_slot = oldest
        self.fir_slot = 0
        self.tf = [0] * TF          # front tail, tf[0] newest
        self.tb = [0] * TB          # back tail, tb[0] newest
        self.a0 = 0                 # S_MFB_A0: partial sum for the next launch's sample 0

    def tail_sum(self):
        return sum(self.taps[t] * self.tb[TB - 1 - t] for t in range(TB))

    def front_eval(self, acc0):
        acc = acc0
        for s in range(TF):
            acc += self.taps[TB + s] * self.tf[TF - 1 - s]
        slot = self.fir_slot
        for s in range(L):
            acc += self.taps[TB + TF + s] * self.lds[slot]
            slot = (slot + 1) % L
        return acc

    def launch(self, x):
        """One kernel launch over the samples x (their B-parts all run): returns the filter outputs y for these samples."""
        nB = len(x)
        acc = [None, None]
        oldx = [None, None]
        # back prologue: sums for samples 0 and 1, the entry arriving during sample 0 = what the front half would have announced
        acc_prev, acc_last = self.a0, self.tail_sum()
        acc[0], acc[1] = acc_prev, acc_last
        xin = self.tf[TF - 2]
        y = []
        # front prologue (behind the extra barrier): sample 0 from the saved history
        if nB > 0:
            y.append(self.front_eval(acc[0]))
        for i in range(nB):
            # ---- front half, iteration i: push x[i], announce, form sample i + 1
            f_acc_slot = (i + 1) & 1
            self.tf = [self.lds[self.fir_slot]] + self.tf[:-1]
            self.lds[self.fir_slot] = x[i]
            self.fir_slot = (self.fir_slot + 1) % L
            announced = self.tf[TF - 2]
            y_next = self.front_eval(acc[f_acc_slot]) if i + 1 < nB else None
            # ---- back half, iteration i (same phase): take the entry announced one iteration ago, sum for sample i + 2
            if i > 0:
                xin = oldx[i & 1]
            self.tb = [xin] + self.tb[:-1]
            acc_prev, acc_last = acc_last, self.tail_sum()
            # ---- barrier: mailbox writes of this phase become visible
            oldx[(i + 1) & 1] = announced
            acc[i & 1] = acc_last
            if y_next is not None:
                y.append(y_next)
        self.a0 = acc_prev
        return y


@pytest.mark.parametrize("firn,l,tb", [(80, 32, 18), (80, 32, 22), (80, 36, 32), (160, 72, 52), (160, 72, 60), (160, 72, 40)])
def test_three_way_history_matches_the_plain_filter(firn, l, tb):
    global FIRN, L, TB, TF
    FIRN, L, TB = firn, l, tb
    TF = FIRN - L - TB
    rng = np.random.default_rng(8)
    taps = [int(v) for v in rng.integers(-9, 10, FIRN)]
    p = Pair(taps)
    hist = [0] * FIRN                      # x[n-80 .. n-1], oldest first
    sizes = [0, 1, 1, 2, 3, 0, 5, 1, 40, 2, 81, 7, 1, 0, 2, 100, 161, 3, 200]
    for n in sizes:
        x = [int(v) for v in rng.integers(-50, 51, n)]
        y = p.launch(x)
        assert len(y) == n
        for i in range(n):
            want = sum(taps[t] * hist[t] for t in range(FIRN))   # the output of a sample does not contain that sample
            assert y[i] == want, (n, i)
            hist = hist[1:] + [x[i]]
