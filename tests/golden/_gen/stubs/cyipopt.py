"""Minimal stand-in for the ``cyipopt`` module, used ONLY by
``tests/golden/_gen/make_golden.py`` inside the build container so that the
reference package (``/root/reference/opty``) can be imported without IPOPT
(``opty/direct_collocation.py:10`` does ``import cyipopt`` unconditionally).

Nothing in the product, the tests run on the GPU box, ``bench.py`` or
``__graft_entry__`` imports this file.
"""


class Problem(object):
    def __init__(self, n=None, m=None, lb=None, ub=None, cl=None, cu=None,
                 **kwargs):
        self._n, self._m = n, m

    def add_option(self, *args, **kwargs):
        pass

    def solve(self, *args, **kwargs):
        raise RuntimeError('IPOPT is not available in this container.')
