"""TEST INFRASTRUCTURE ONLY: restatement of the reference's Poisson approximation
(moira/moira.py:1637-1679 calculate_errors_poisson, :1723-1733 interpolate), used by tests/ as the
checker of the GPU lambda reduction + host tail (mpb_filter_poisson_host).  Pinned by the reference's
own KAT (moira/test/test_moira.py:45: 6.932519986616133).  Nothing under moira_amd/ imports this."""
import math


def interpolate(e1, p1, e2, p2, alpha):
    r = e1 + ((e2 - e1) * ((1 - alpha) - p1) / (p2 - p1))          # moira.py:1729
    return 0 if r < 0 else r


def calculate_errors_poisson(sequence, quals, alpha):
    lam, ns = 0, 0
    for base, q in zip(sequence, quals):                            # moira.py:1656-1663
        if q < 0:
            raise ValueError("Qualities must have positive values.")
        if base == "N":
            ns += 1
        else:
            lam += 10 ** (q / -10.0)
    acc, j = [0], 0
    while True:                                                     # moira.py:1667-1676
        acc.append(acc[-1] + (math.exp(-lam) * (lam ** j)) / math.factorial(j))
        if acc[-1] > (1 - alpha):
            break
        j += 1
    return interpolate(j - 1, acc[-2], j, acc[-1], alpha), ns
