"""Per-process overrides of choices the library normally takes itself - which fused kernel route a run takes, the tile
geometry, whether runs are captured as hipGraphs.  The defaults ARE the product; tests set attributes of ``HINTS`` to pin
both kernel routes against the reference, development tools to measure them.  Nothing in the package reads the
environment: ``apply_mapping`` exists so that *test / tool* infrastructure can translate a mapping it owns (its own
command line, the ``PF_*`` variables of a test driver's subprocess) - the package never calls it.

The kernel-side choices travel to the library per call in ``pf_filter_args.hints`` (``include/pf_amd.h: pf_run_hints``)."""

ROUTE_AUTO, ROUTE_PER_STEP, ROUTE_COLUMN_GENERIC, ROUTE_CLUSTER, ROUTE_CLUSTER_ALWAYS, ROUTE_CLUSTER_SPREAD = 0, 1, 2, 3, 4, 5


class RunHints:
    __slots__ = ("route", "column_max_n", "tile_target", "ancestor_search", "fused_step", "fused_batch", "graph", "direct",
                 "theta_kernels", "cluster", "cluster_patience")

    def __init__(self):
        self.reset()

    def reset(self):
        self.route = ROUTE_AUTO      # pf_run_hints.route
        self.cluster = True          # ROUTE_AUTO travels as PF_ROUTE_CLUSTER: self-contained runs of filters of 2 049 .. 16 384
        #                              particles take the column-cluster kernel.  Opt-in at the C ABI because its workgroups wait
        #                              for each other and a launch that cannot make progress REPORTS it (pf_filter_args.status)
        #                              instead of hanging: this package passes the status word with every such run, reads it
        #                              where it next waits for the device and re-issues the piece on the per-step route - same
        #                              draws, same numbers (filters/particle/base.py: _verified_block, inference/smc2.py).  Runs on
        #                              several streams / threads / processes of one device share the slots (pf_cluster.hpp)
        self.cluster_patience = 0    # pf_run_hints.cluster_patience (0 = the library's 2^21 polls; tests: -1 forces the give-up path)
        self.column_max_n = 0        # pf_run_hints.column_max_n (0 = the library's 2048)
        self.tile_target = 0         # pf_run_hints.tile_target (0 = the library's 1024 workgroups per launch)
        self.ancestor_search = 0     # pf_run_hints.ancestor_search
        self.fused_step = True       # ``filter()`` of a built-in model takes the fused single-step move
        self.fused_batch = True      # ``batch_filter()`` of a built-in model takes the fused run
        self.graph = True            # repeated fused runs of one configuration replay a captured hipGraph
        self.theta_kernels = True    # SMC^2 / PMMH moves of scalar standard-family priors: theta arithmetic in pf_theta_* (else torch)
        self.direct = False          # every plain fused run takes the direct driver (no persistent plan, no graph), not only single-launch runs

    def key(self):
        """What a cached launch plan depends on."""
        return (self.kernel_route(), self.column_max_n, self.tile_target, self.ancestor_search, self.cluster_patience)

    def kernel_route(self):
        """``pf_run_hints.route`` of the next call."""
        return ROUTE_CLUSTER if (self.route == ROUTE_AUTO and self.cluster) else self.route

    def cluster_takes(self, n, b, resampler_systematic=True):
        """Does a self-contained run of ``b`` filters of ``n`` particles take the column-cluster kernel (the library's rule,
        ``pf_kernels.hip: cluster_eligible``)?  One launch per run - or two - with nothing for a hipGraph to replay."""
        route = self.kernel_route()
        if route not in (ROUTE_CLUSTER, ROUTE_CLUSTER_ALWAYS, ROUTE_CLUSTER_SPREAD) or not resampler_systematic:
            return False
        if n <= max(2048, self.column_max_n or 2048) or n > 16384 or n % 4:
            return False
        members = ((n + 1023) // 1024) * b
        return members <= 2048 if route == ROUTE_CLUSTER else members <= 8192  # (beyond: no record space is reserved)

    def fill(self, args):
        """Writes the kernel-side choices into a ``PfFilterArgs``."""
        h = args.hints
        h.route, h.column_max_n, h.tile_target, h.ancestor_search = self.kernel_route(), self.column_max_n, self.tile_target, self.ancestor_search
        h.cluster_patience = self.cluster_patience
        h.cluster_generation = 0  # (per launch: the drivers that own a zero-filled workspace number their cluster launches)
        h.resume = h.prepare_next = 0  # (per-call facts, set by the move loops that know them)

    def apply_mapping(self, m):
        """``PF_NO_COLUMN / PF_COLUMN_GENERIC / PF_CLUSTER (always) / PF_NO_CLUSTER / PF_COLUMN_MAX_N / PF_TARGET_WGS / PF_FORCE_SEARCH / PF_NO_FUSED_STEP /
        PF_NO_FUSED_BATCH / PF_NO_GRAPH / PF_DIRECT / PF_NO_THETA_KERNELS`` of a mapping the CALLER owns -> attributes (absent keys: the defaults)."""
        on = lambda k: str(m.get(k, "0")) not in ("", "0")  # noqa: E731
        self.route = ROUTE_PER_STEP if on("PF_NO_COLUMN") else (ROUTE_COLUMN_GENERIC if on("PF_COLUMN_GENERIC") else
                                                                (ROUTE_CLUSTER_ALWAYS if on("PF_CLUSTER") else ROUTE_AUTO))
        self.cluster = not on("PF_NO_CLUSTER")
        self.column_max_n = int(m.get("PF_COLUMN_MAX_N", 0) or 0)
        self.tile_target = int(m.get("PF_TARGET_WGS", 0) or 0)
        self.ancestor_search = 1 if on("PF_FORCE_SEARCH") else 0
        self.fused_step, self.fused_batch, self.graph = not on("PF_NO_FUSED_STEP"), not on("PF_NO_FUSED_BATCH"), not on("PF_NO_GRAPH")
        self.direct = on("PF_DIRECT")
        self.theta_kernels = not on("PF_NO_THETA_KERNELS")
        self.cluster_patience = int(m.get("PF_CLUSTER_PATIENCE", 0) or 0)
        return self


HINTS = RunHints()
