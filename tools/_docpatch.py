"""Development helper: replace(text, old, new) / between(text, start, end, new) that refuse to act
unless the anchor occurs exactly once (an empty or missing anchor must never reach str.replace)."""
def replace(s, old, new):
    assert old and s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)
def between(s, start, end, new):
    a = s.index(start)
    b = s.index(end, a + len(start))
    assert 0 <= a < b
    return s[:a] + new + s[b:]
