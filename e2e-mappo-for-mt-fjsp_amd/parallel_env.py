"""Parallel_env — drop-in for the reference's trainer/parallel_env.py::Parallel_env (pe:19-282).

Same constructor argument (the `args` dict), same method names, same numpy return shapes and dtypes, same
`paral_env_DG[i]` / `paral_Rscaling_instance[i]` members that Run.py, algorithm/ppo_algorithm.py and
trainer/validate.py touch — but every instance lives in HBM and every method is one HIP kernel
launch through libmtfjsp.so (include/mtfjsp.h).  Use DeviceBatchEnv directly for the zero-copy rollout; this
class is the compatibility surface for unmodified reference callers (it copies observations to host numpy,
like the reference returns them).

Differences that are deliberate (documented in DESIGN.md §6):
  * an action that is already scheduled / whose job predecessor is unscheduled raises ValueError instead of
    silently corrupting node attributes (env:1496-1528); never reached under the reference's masks;
  * `paral_env_DG[i].render()` prints a console Gantt chart (the reference's default "gantt_console" visualisation,
    Run.py:651-653, validate.py:286); window / rgb-array rendering is outside the accelerated hot path;
  * `paral_env_DG[i].step([task, machine])` (gym-style, env:716-974) steps instance i alone through the same kernel (the other
    instances of the batch receive a rejected no-op) and returns the reference's 14-tuple; the ONE entry left None is `state`
    (entry 0 of step's and reset's tuples: the flat normalised [T, T + M + 1] legacy observation of env:2073-2140, which no caller
    in Run.py / validate.py / ppo_algorithm.py reads).  ft_s (finish times of the scheduled tasks, env:2146-2151), it_s (idle time
    each decision added, env:2150) and the 3-column tasks_fea (env:2215-2241) are filled, bit-equal to the reference's;
    `observation_space` / `action_space` (env:434-467, the defaults Parallel_env constructs its environments with: task actions,
    normalised flat float32 observation) are gym / gymnasium spaces when either package is importable and objects with the same
    `shape`, `n`, `low`, `high`, `dtype` attributes otherwise.
`DisjunctiveGraphJspEnv_singleStep` below is the same surface as a stand-alone one-instance environment for callers that
build their own env (trainer/validate.py:108-127).
"""

import numpy as np
import torch

from . import capi
from .batch_env import DeviceBatchEnv
from .instances import random_weights


class _Space:
    """what callers read of a gym space (shape / n / low / high / dtype / contains / sample) when neither gym nor gymnasium is installed"""

    def __init__(self, n=None, shape=None, low=0.0, high=1.0, dtype=np.float32):
        self.n, self.low, self.high, self.dtype = n, low, high, np.dtype(dtype if n is None else np.int64)
        self.shape = () if n is not None else tuple(shape)

    def contains(self, x):
        if self.n is not None:
            return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def sample(self):
        return int(np.random.randint(self.n)) if self.n is not None else np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    def __repr__(self):
        return f"Discrete({self.n})" if self.n is not None else f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


def _spaces(T, M):
    """(observation_space, action_space) of env:434-467 with the constructor defaults (action_mode 'task', normalised flat float32)"""
    shape = (T * (T + M + 1),)
    for mod in ("gymnasium", "gym"):
        try:
            sp = __import__(mod).spaces
            return sp.Box(low=0.0, high=1.0, shape=shape, dtype=np.float32), sp.Discrete(T)
        except Exception:
            continue
    return _Space(shape=shape), _Space(n=T)


def _tuple_extras(tfea, finish, it_s):
    """ft_s (env:2146-2151), it_s (env:2150), the 3-column tasks_fea (env:2215-2241) of one instance from its [T,12] feature rows
    (col 1 estimated / real finish, col 2 energy, col 3 scheduled), its finish times (NaN = unscheduled) and the idle-time record"""
    sched = tfea[:, 3] == 1
    ft_s = np.where(sched, np.nan_to_num(finish, nan=0.0), 0.0)
    old = np.stack([tfea[:, 1], np.where(sched, tfea[:, 2], 0.0), tfea[:, 3]], axis=1)
    return ft_s, np.array(it_s, dtype=np.int64), old


class _NodeView:
    """G.nodes[k] -> dict with the attributes callers read (ppo:271-273, validate.py:142,222)."""

    def __init__(self, proxy):
        self._p = proxy

    def __getitem__(self, k):
        p = self._p
        T = p._env.T
        if k == 0 or k == T + 1:
            return {"machine": -2, "duration": 0, "scheduled": k == 0, "start_time": 0 if k == 0 else None,
                    "finish_time": 0 if k == 0 else None, "job": -1}
        a = k - 1
        par = p._parent
        mach = int(par._mirror(capi.STATE_MACHINE)[p._i, a])
        sched = mach >= 0
        return {
            "machine": mach if sched else -1,
            "scheduled": sched,
            "start_time": float(par._mirror(capi.STATE_START)[p._i, a]) if sched else None,
            "finish_time": float(par._mirror(capi.STATE_FINISH)[p._i, a]) if sched else None,
            "duration": float(par.ability_instance[p._i][0][a, mach]) if sched else 0,
            "job": a // p._env.M,
        }


class _GraphView:
    def __init__(self, proxy):
        self.nodes = _NodeView(proxy)


class _EnvProxy:
    """Stand-in for one DisjunctiveGraphJspEnv_singleStep inside `paral_env_DG` (read-only view of instance i)."""

    def __init__(self, parent, i):
        self._parent, self._i, self._env = parent, i, parent._dev
        self.G = _GraphView(self)
        self.observation_space, self.action_space = parent._spaces

    @property
    def reward_random_weight(self):
        return self._parent._w3[self._i].copy()

    def _prev(self, c):
        return float(self._parent._mirror(capi.STATE_PREV_COSTS)[self._i, c])

    makespan_previous_step = property(lambda s: s._prev(0))
    total_e1_previous_step = property(lambda s: s._prev(1))
    trans_t_previous_step = property(lambda s: s._prev(2))
    idle_t_previous_step = property(lambda s: s._prev(3))

    @property
    def machine_routes(self):
        r = self._parent._mirror(capi.STATE_ROUTES)[self._i]
        return {m: (r[m][r[m] >= 0] + 1).astype(np.float64) for m in range(self._env.M)}   # node ids, like the reference

    def valid_action_mask(self, action_mode=None):
        return [bool(x) for x in self._parent._valid_mask()[self._i]]

    def reset(self, Random_weight_type="01"):
        """env.reset() (env:1183-1245) as Run.py:660 calls it after `done`: consumes the same three
        `random.uniform` draws; the instance itself is re-initialised by the next init_DGFJSPEnv_state0()."""
        w = random_weights(1, kind=Random_weight_type, config_weights=self._parent._w_cfg)[0]
        self._parent._w3[self._i] = w
        return None

    def step(self, joint_action):
        """env.step([task_idx, m_idx]) (env:716-974) for THIS instance only -> the reference's 14-tuple
        (state, reward, done, info, r_mk, r_idle, r_pt, r_tt, ft_s, it_s, adj_wrk, tasks_fea, machine_fea, tasks_fea_1101);
        rewards are the unscaled ones, as env.step returns them (reward scaling is Parallel_env's, pe:255-260)."""
        return self._parent._step_one(self._i, int(joint_action[0]), int(joint_action[1]))

    def render(self, mode="human", show=None, **render_kwargs):
        """console Gantt chart of the current (partial) schedule — what the reference prints for its default
        "gantt_console" visualisation (env:1274-1354 -> visualizer); returns None like mode="human"."""
        par, i = self._parent, self._i
        routes = par._mirror(capi.STATE_ROUTES)[i]
        st, ft = par._mirror(capi.STATE_START)[i], par._mirror(capi.STATE_FINISH)[i]
        M = self._env.M
        horizon = float(np.nanmax(ft)) if np.isfinite(ft).any() else 0.0
        width = 60
        print(f"Gantt (instance {i}, makespan so far {horizon:.1f})")
        for m in range(M):
            line = [" "] * width
            for a in routes[m][routes[m] >= 0]:
                lo = int(st[a] / horizon * (width - 1)) if horizon > 0 else 0
                hi = max(lo, int(ft[a] / horizon * (width - 1)) if horizon > 0 else 0)
                ch = "0123456789abcdefghijklmnopqrstuvwxyz"[(int(a) // M) % 36]          # one symbol per job
                for x in range(lo, hi + 1):
                    line[x] = ch
            print(f"  machine {m:2d} |" + "".join(line) + "|")
        return None


class _ScalerProxy:
    """paral_Rscaling_instance[i] — only .reset() (zero the discounted return R, pt:123) is used by callers."""

    def __init__(self, parent, i):
        self._parent, self._i = parent, i

    def reset(self):
        self._parent._scaler_pending[self._i] = True


class Parallel_env(object):
    def __init__(self, args):
        self.njobs = args['n_job']
        self.nmachines = args['n_machine']
        self.ntasks = self.njobs * self.nmachines
        self.nedges = args['n_edge']
        self.batch_size = args['env_batch']
        self.m_scaling = args['m_scaling']
        self.reward_dict = args['reward_scaling']
        self.args = args
        self.ability_instance = []
        self.paral_Rscaling_instance = []
        self.paral_env_DG = []
        self.oenv_info = []
        self._w_cfg = (args['weight_mk'], args['weight_ec'], args['weight_tt'])
        self._left_shift = bool(args.get('perform_left_shift_if_possible', True))
        self._device = int(args.get('hip_device', 0))
        self._dev = DeviceBatchEnv(self.njobs, self.nmachines, self.nedges, self.batch_size, left_shift=self._left_shift,
                                   obs_dtype="f64", device=self._device, gamma=args['GAMMA'], w_cfg=self._w_cfg,
                                   scaling_divisor=self.reward_dict['scaling_divisor'])
        self._w3 = np.zeros((self.batch_size, 3))
        self._cache = {}
        self._scaler_pending = np.zeros(self.batch_size, bool)
        # env.it_s: the idle time each decision added (env:2150), cleared by reset — an INTEGER array in the reference (reset's
        # _state_array turns the list of zeros into an int64 array; every later assignment truncates toward zero), and so here
        self._it_s = np.zeros((self.batch_size, self.ntasks), np.int64)
        self._spaces = _spaces(self.ntasks, self.nmachines)

    # -- host mirrors of device state, refreshed lazily once per step
    def _mirror(self, which):
        if which not in self._cache:
            self._cache[which] = self._dev.read_state(which)
        return self._cache[which]

    def _valid_mask(self):
        if "vmask" not in self._cache:
            self._cache["vmask"] = self._dev.valid_action_mask().cpu().numpy()
        return self._cache["vmask"]

    def _flush_scaler_resets(self):
        if self._scaler_pending.any():
            if self._scaler_pending.all():
                self._dev.scaler_reset_returns()
            else:
                self._dev.scaler_reset_returns_masked(self._scaler_pending)
            self._scaler_pending[:] = False

    @staticmethod
    def _np(x):
        return x.numpy() if torch.is_tensor(x) else np.asarray(x)

    def get_batch(self, dataset_dict):
        """pe:39-66"""
        t = self._np(dataset_dict["t"]).astype(np.float64)
        p = self._np(dataset_dict["p"]).astype(np.float64)
        tt = self._np(dataset_dict["transT"]).astype(np.float64)
        edge = self._np(dataset_dict["edge"])
        B = self.batch_size
        self.ability_instance = [[t[i].copy(), p[i].copy(), tt[i].copy(), edge[i].copy()] for i in range(B)]
        self._dev.load_instances(t[:B], p[:B], tt[:B], edge=edge[:B])

    def init_RewardScaling_sameBATCH(self, shape):
        """pe:70-85"""
        assert shape == 4
        self._dev.scaler_init()
        self._scaler_pending[:] = False
        self.paral_Rscaling_instance = [_ScalerProxy(self, i) for i in range(self.batch_size)]

    def _host_obs(self):
        d = self._dev
        adj = d.dense_adj().cpu().numpy()
        return adj, d.m_fea2.cpu().numpy().copy(), d.tasks_fea.cpu().numpy().copy()

    def init_DGFJSPEnv_state0(self):
        """pe:87-149 -> (adj [B,T,T], m_fea2 [B,M,8], tasks_fea [B*T,12]) float64"""
        self._w3 = random_weights(self.batch_size, kind="01", config_weights=self._w_cfg)   # pe:130 calls env.reset() with its default type "01" (env:1183)
        self._dev.reset(self._w3)
        self._cache = {}
        self._it_s[:] = 0
        self.paral_env_DG = [_EnvProxy(self, i) for i in range(self.batch_size)]
        return self._host_obs()

    def cal_cur_task_machine_feature(self, task_index, m_mask, all_task_fea):
        """pe:152-214 -> ndarray [B,M,6] float64.  `all_task_fea` is accepted for signature compatibility; the
        predecessor's machine is read from device state, which equals all_task_fea[a-1][5]-1 for the observation
        returned by the latest reset/step (the only way Run.py:347 and validate.py call it)."""
        ti = self._np(task_index.cpu() if torch.is_tensor(task_index) else task_index).astype(np.int32).reshape(-1)
        mm = self._np(m_mask.cpu() if torch.is_tensor(m_mask) else m_mask).reshape(self.batch_size, self.nmachines).astype(np.uint8)
        out = self._dev.observe_mfea1(ti, mm)
        return out.cpu().numpy().copy()

    def DGFJSPEnv_paral_step(self, joint_actions):
        """pe:217-268 -> (adj_, oenv_info, m_fea2_, tasks_fea_)"""
        a = np.array([ja[0] for ja in joint_actions], np.int32)
        m = np.array([ja[1] for ja in joint_actions], np.int32)
        self._flush_scaler_resets()
        try:
            self._dev.step(a, m)
        except capi.MtfjspError as e:
            if e.code == capi.ERR_ACTION:
                raise ValueError(str(e)) from None
            raise
        self._cache = {}
        status = self._dev.status.cpu().numpy()
        for l in np.flatnonzero(status & capi.ST_INFEASIBLE):          # pe:246-248
            print("!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!!")
            print(f"============= 'DGFJSPEnv_paral_step' occur error: chose Minus: t={self.ability_instance[l][0][a[l]][m[l]]}, p= {self.ability_instance[l][1][a[l]][m[l]]}")
        info = self._dev.info.cpu().numpy()
        ok = (status & capi.ST_INVALID) == 0
        self._it_s[np.flatnonzero(ok), a[ok]] = -self._dev.raw.cpu().numpy()[ok, 2]     # idle_this - idle_prev = -(r_idle) exactly (env:1088, 2150)
        self.oenv_info = [[info[i, 0], bool(info[i, 1]), info[i, 2], info[i, 3], info[i, 4], info[i, 5]]
                          for i in range(self.batch_size)]
        adj, mfea2, tfea = self._host_obs()
        return adj, self.oenv_info, mfea2, tfea

    def _step_one(self, i, a, m):
        """per-env gym step of instance i (proxy.step / DisjunctiveGraphJspEnv_singleStep.step, env:716-974): every other instance
        gets task index -1, which the kernel rejects leaving its STATE untouched (MTFJSP_ST_INVALID; its info / raw reward rows of
        the last batched step are cleared).  The fused step kernel also applies RewardScaling, which the reference's env.step never
        does (only the batched step, pe:255-260): instance i's scaler state is saved before and restored after the launch, so
        mixing proxy steps with DGFJSPEnv_paral_step leaves the scaled rewards exactly the reference's."""
        B, T, M = self.batch_size, self.ntasks, self.nmachines
        ta = torch.full((B,), -1, dtype=torch.int32); ma = torch.zeros(B, dtype=torch.int32)
        ta[i], ma[i] = a, m
        self._flush_scaler_resets()
        scaler_i = self._dev.read_state(capi.STATE_SCALER)[i].copy()
        self._dev.step(ta.to(self._dev.device), ma.to(self._dev.device))
        self._dev.set_scaler_state(i, scaler_i)
        self._cache = {}
        status = int(self._dev.status[i].item())
        if status & capi.ST_INVALID:
            raise ValueError(f"invalid action for instance {i}: task {a} is scheduled already, its job predecessor is not, or an index is out of range")
        raw = self._dev.raw[i].cpu().numpy()
        done = bool(self._dev.info[i, 1].item())
        adj = self._dense_adj_row(i)
        tfea = self._dev.tasks_fea.view(B, T, 12)[i].cpu().numpy().copy()
        mfea = self._dev.m_fea2[i].cpu().numpy().copy()
        self._it_s[i, a] = -float(raw[2])                           # env:2150
        ft_s, it_s, tfea3 = _tuple_extras(tfea, self._mirror(capi.STATE_FINISH)[i], self._it_s[i])
        return (None, float(raw[0]), done, {}, float(raw[1]), float(raw[2]), float(raw[3]), float(raw[4]), ft_s, it_s,
                adj, tfea3, mfea, tfea)

    def _dense_adj_row(self, i):
        """dense adj_wrk [T,T] of ONE instance from its 2 T ELL entries (row = destination, self loop 1; env:2019-2073) — not the
        whole [B,T,T] export"""
        T = self.ntasks
        col = self._dev.ell_col.view(self.batch_size, T, 2)[i].cpu().numpy()
        val = self._dev.ell_val.view(self.batch_size, T, 2)[i].cpu().numpy()
        adj = np.eye(T)
        for k in range(2):
            ok = col[:, k] >= 0
            adj[np.flatnonzero(ok), col[ok, k]] = val[ok, k]
        return adj

    def reset_data(self):
        """pe:271-282"""
        self.paral_env_DG = []
        self.oenv_info = []


class DisjunctiveGraphJspEnv_singleStep:
    """One MT-FJSP instance with the gym-style surface of the reference's env class (env:34-368: constructor keywords of
    pe:108-118 / validate.py:108-127; reset env:1183-1245; step env:716-974; valid_action_mask env:2535-2575; render) on top
    of a one-instance device batch.  For the B = 1 callers that construct their own env (trainer/validate.py); a training
    rollout should use Parallel_env / DeviceBatchEnv, which step thousands of instances per launch."""

    def __init__(self, jps_instance=None, ability_tr_mm=None, reward_function='wrk', reward_function_parameters=None,
                 perform_left_shift_if_possible=True, default_visualisations=None, configs=None, edge=None, **_ignored):
        t, p = np.asarray(jps_instance[0], np.float64), np.asarray(jps_instance[1], np.float64)
        cfg = dict(configs) if configs is not None else {}
        M = t.shape[1]
        J = t.shape[0] // M
        E = int(cfg.get('n_edge', 1))
        if reward_function != 'wrk':
            raise ValueError("only the 'wrk' reward function of the training / evaluation path is provided (env:1051-1171)")
        args = {'n_job': J, 'n_machine': M, 'n_edge': E, 'env_batch': 1, 'm_scaling': cfg.get('m_scaling', 1),
                'reward_scaling': reward_function_parameters or {'scaling_divisor': 1}, 'GAMMA': cfg.get('GAMMA', 0.99),
                'weight_mk': cfg.get('weight_mk', 0.4), 'weight_ec': cfg.get('weight_ec', 0.4), 'weight_tt': cfg.get('weight_tt', 0.2),
                'perform_left_shift_if_possible': perform_left_shift_if_possible, 'hip_device': cfg.get('hip_device', 0)}
        self._pe = Parallel_env(args)
        if edge is None:                                              # machines split evenly into shops in index order (generate…py:219-233)
            edge = np.arange(M).reshape(E, M // E)
        self._pe.get_batch({"t": t[None], "p": p[None], "transT": np.asarray(ability_tr_mm, np.float64)[None], "edge": np.asarray(edge)[None]})
        self._pe.init_RewardScaling_sameBATCH(4)
        self.n_jobs, self.n_machines, self.total_tasks_without_dummies = J, M, J * M
        self.observation_space, self.action_space = self._pe._spaces
        self._proxy = None

    def reset(self, Random_weight_type="01"):
        """-> the 9-tuple of env._state_array (env:2515): (state, ft_s, it_s, adj_wrk, tasks_fea, machines_fea, tasks_fea_1101,
        ft_estimated [T], pt_estimated [T]); `state` (the flat legacy observation, read by no caller) is None"""
        pe = self._pe
        pe._w3 = random_weights(1, kind=Random_weight_type, config_weights=pe._w_cfg)
        pe._dev.reset(pe._w3)
        pe._cache = {}
        pe._it_s[:] = 0
        pe.paral_env_DG = [_EnvProxy(pe, 0)]
        self._proxy = pe.paral_env_DG[0]
        adj, mfea2, tfea = pe._host_obs()
        ft_s, it_s, tfea3 = _tuple_extras(tfea, np.full(tfea.shape[0], np.nan), pe._it_s[0])
        return (None, ft_s, it_s, adj[0], tfea3, mfea2[0], tfea, tfea[:, 1].copy(), tfea[:, 2].copy())

    def step(self, joint_action):
        return self._proxy.step(joint_action)

    def valid_action_mask(self, action_mode=None):
        return self._proxy.valid_action_mask(action_mode)

    def render(self, *a, **k):
        return self._proxy.render(*a, **k)

    def __getattr__(self, name):      # G, machine_routes, reward_random_weight, *_previous_step: the proxy's read-only views
        if name.startswith("_") or self.__dict__.get("_proxy") is None:
            raise AttributeError(name)
        return getattr(self._proxy, name)
