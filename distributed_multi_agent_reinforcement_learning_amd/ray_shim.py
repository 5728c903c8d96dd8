"""The handful of `ray` names the reference's driver uses, on plain Python objects -- SURVEY 8b's "thin `.remote` / `ray.get` shim".

The reference's training loop (main.py:79-158) is written against Ray actors: `Learner.options(...).remote(cfg, ...)`,
`worker.run.remote(w_a, w_c)`, `ray.get(ref)`, `ray.wait(refs, num_returns=1, timeout=0.1)`, `ray.put(obj)`.  One process drives
one GPU here, so an "actor" is the object itself and a "remote call" is a direct call whose result is wrapped in an `ObjectRef`;
with this module bound to the name `ray` and `remote(Learner)`, `remote(Worker)`, `remote(EvaluatorProc)` bound to the class
names, that loop text runs unchanged (tests/test_runner_gpu.py runs it).

    import distributed_multi_agent_reinforcement_learning_amd.ray_shim as ray
    Learner = ray.remote(runner.Learner); Worker = ray.remote(runner.Worker); EvaluatorProc = ray.remote(evaluator.EvaluatorProc)

Semantics kept from Ray, because the loop relies on them:
* arguments that are `ObjectRef`s (also inside a list / tuple argument, as in `collect_buffer.remote(worker_run_ref[node])`) are
  resolved before the call; an exception inside a call surfaces at `get`, not at `.remote()`;
* an actor executes its calls one at a time, in submission order;
* `.options(background=True)` gives the actor a worker thread and a HIP stream of its own: its calls return at once and run
  concurrently with the caller -- the reference's evaluator is such an actor (main.py:135-158 polls it with `ray.wait(...,
  timeout=0.1)` and keeps training).  Every other `.options(...)` keyword (resources, num_cpus, ...) is accepted and ignored.
"""
import queue
import threading
import time

_UNSET = object()


class ObjectRef:
    """Result of a remote call (or of `put`): a value, an exception, or a call still running on an actor's thread."""

    __slots__ = ("_value", "_error", "_done")

    def __init__(self, value=_UNSET):
        self._value = value
        self._error = None
        self._done = threading.Event()
        if value is not _UNSET:
            self._done.set()

    def _finish(self, value=None, error=None):
        self._value, self._error = value, error
        self._done.set()

    def ready(self):
        return self._done.is_set()

    def result(self, timeout=None):
        if not self._done.wait(timeout):
            raise TimeoutError("remote call still running")
        if self._error is not None:
            raise self._error
        return self._value


def _resolve(arg):
    """Ray hands a task the VALUE of an ObjectRef argument; one level of list / tuple nesting is resolved too (the reference passes
    lists of refs where Ray code would call ray.get on them inside the task)."""
    if isinstance(arg, ObjectRef):
        return arg.result()
    if isinstance(arg, (list, tuple)) and any(isinstance(a, ObjectRef) for a in arg):
        return type(arg)(a.result() if isinstance(a, ObjectRef) else a for a in arg)
    return arg


def put(value):
    return ObjectRef(value)


def get(refs, timeout=None):
    if isinstance(refs, (list, tuple)):
        return [get(r, timeout) for r in refs]
    return refs.result(timeout) if isinstance(refs, ObjectRef) else refs


def wait(object_refs, num_returns=1, timeout=None):
    """-> (ready, not_ready), at most `num_returns` ready refs, waiting up to `timeout` seconds for them (ray.wait)."""
    object_refs = list(object_refs)
    deadline = None if timeout is None else time.monotonic() + timeout
    while True:
        ready = [r for r in object_refs if not isinstance(r, ObjectRef) or r.ready()][:num_returns]
        if len(ready) >= min(num_returns, len(object_refs)) or (deadline is not None and time.monotonic() >= deadline):
            rest = [r for r in object_refs if not any(r is x for x in ready)]
            return ready, rest
        time.sleep(0.002)


class _Method:
    def __init__(self, handle, name):
        self._handle, self._name = handle, name

    def remote(self, *args, **kwargs):
        return self._handle._submit(self._name, args, kwargs)

    def __call__(self, *args, **kwargs):
        raise TypeError(f"actor methods are called with .remote(): {self._name}.remote(...)")


class ActorHandle:
    def __init__(self, cls, args, kwargs, background):
        self._background = bool(background)
        self._stream = None
        self._device = None
        if self._background:
            try:
                import torch
                if torch.cuda.is_available():
                    self._device = torch.cuda.current_device()   # the actor works on its creator's GPU
            except ImportError:
                pass
            self._q = queue.Queue()
            self._thread = threading.Thread(target=self._loop, name=f"actor-{cls.__name__}", daemon=True)
            ref = ObjectRef()
            self._q.put(("__init__", (cls, args, kwargs), {}, ref))
            self._thread.start()
            ref.result()   # construction errors surface here, like a failed actor start
        else:
            self._obj = cls(*[_resolve(a) for a in args], **{k: _resolve(v) for k, v in kwargs.items()})

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return _Method(self, name)

    def _call(self, name, args, kwargs):
        if name == "__init__":
            cls, a, k = args
            self._obj = cls(*[_resolve(x) for x in a], **{kk: _resolve(v) for kk, v in k.items()})
            return None
        return getattr(self._obj, name)(*[_resolve(a) for a in args], **{k: _resolve(v) for k, v in kwargs.items()})

    def _submit(self, name, args, kwargs):
        if not self._background:
            ref = ObjectRef()
            try:
                ref._finish(self._call(name, args, kwargs))
            except Exception as e:  # noqa: BLE001 -- delivered at get(), as Ray does
                ref._finish(error=e)
            return ref
        ref = ObjectRef()
        ev = None
        try:
            import torch
            if torch.cuda.is_available():   # the call's device inputs (e.g. a weights snapshot) were produced on the caller's stream
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
        except ImportError:
            pass
        self._q.put((name, args, kwargs, ref, ev))
        return ref

    def _loop(self):
        stream_ctx = None
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(self._device)
                self._stream = torch.cuda.Stream()
                stream_ctx = torch.cuda.stream(self._stream)
                stream_ctx.__enter__()
        except ImportError:
            pass
        while True:
            item = self._q.get()
            if item is None:
                break
            name, args, kwargs, ref = item[:4]
            ev = item[4] if len(item) > 4 else None
            try:
                if ev is not None and self._stream is not None:
                    self._stream.wait_event(ev)
                out = self._call(name, args, kwargs)
                if self._stream is not None:
                    self._stream.synchronize()   # a finished ref means finished device work (the caller may read it on any stream)
                ref._finish(out)
            except Exception as e:  # noqa: BLE001
                ref._finish(error=e)
        if stream_ctx is not None:
            stream_ctx.__exit__(None, None, None)

    def _shutdown(self):
        if self._background:
            self._q.put(None)
            self._thread.join()


class ActorClass:
    def __init__(self, cls, background=False):
        self._cls, self._background = cls, background
        self.__name__ = getattr(cls, "__name__", "Actor")

    def options(self, background=None, **_ignored):
        return ActorClass(self._cls, self._background if background is None else background)

    def remote(self, *args, **kwargs):
        return ActorHandle(self._cls, args, kwargs, self._background)


class _RemoteFunction:
    def __init__(self, fn):
        self._fn = fn

    def options(self, **_ignored):
        return self

    def remote(self, *args, **kwargs):
        ref = ObjectRef()
        try:
            ref._finish(self._fn(*[_resolve(a) for a in args], **{k: _resolve(v) for k, v in kwargs.items()}))
        except Exception as e:  # noqa: BLE001
            ref._finish(error=e)
        return ref


def remote(*args, **kwargs):
    """@ray.remote / @ray.remote(num_cpus=1, num_gpus=0.001) on a class or a function."""
    if len(args) == 1 and not kwargs and (isinstance(args[0], type) or callable(args[0])):
        target = args[0]
        return ActorClass(target) if isinstance(target, type) else _RemoteFunction(target)
    return lambda target: ActorClass(target) if isinstance(target, type) else _RemoteFunction(target)


def init(*_a, **_k):
    return None


def shutdown():
    return None


def is_initialized():
    return True
