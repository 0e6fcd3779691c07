"""Interactive generation: the session object `run_interactive_generation` drives, and its value types.

Mirrors the public surface of the reference's `inferix/core/interactive/session.py:38-560` (InteractiveSession) and
`inferix/core/types/interactive.py` (enums, QueuedInput, GenerationStatus, CheckpointResult, SegmentBoundary and the two
helpers) — the names and fields callers bind to — for the single-process, one-rank-per-GPU case: a UI thread submits prompt /
guidance / control inputs at any time, the generation thread picks up the LATEST queued one at the next checkpoint (segment
boundary under NEXT_SEGMENT, block boundary under NEXT_BLOCK), pause / resume / stop act through an event and a flag.
The reference's worker broadcast of an applied input (`_broadcast_input_to_workers`, :461-520, torch.distributed object
broadcast from rank 0) is the control plane's business and is not reproduced: under sequence parallelism every rank runs
the same session object fed by the same queue.
"""
from __future__ import annotations

import threading
import time
import uuid
from dataclasses import dataclass, field
from enum import Enum
from typing import Callable, List, Optional


class InputApplyPolicy(Enum):
    NEXT_SEGMENT = "next_segment"
    NEXT_BLOCK = "next_block"


class InputState(Enum):
    QUEUED = "queued"
    PENDING = "pending"
    APPLIED = "applied"
    DISCARDED = "discarded"


class SessionState(Enum):
    IDLE = "idle"
    GENERATING = "generating"
    PAUSED = "paused"
    COMPLETED = "completed"
    ERROR = "error"


class ControlCommand(Enum):
    CONTINUE = "continue"
    PAUSE = "pause"
    RESUME = "resume"
    STOP = "stop"
    MODIFY_PARAMS = "modify"


@dataclass
class QueuedInput:
    input_id: str
    prompt: Optional[str] = None
    guidance_scale: Optional[float] = None
    control: ControlCommand = ControlCommand.CONTINUE
    state: InputState = InputState.QUEUED
    apply_policy: InputApplyPolicy = InputApplyPolicy.NEXT_SEGMENT
    estimated_wait_seconds: float = 0.0
    will_apply_at: str = ""
    queued_at: float = 0.0

    def to_dict(self) -> dict:
        return {"input_id": self.input_id, "prompt": self.prompt, "guidance_scale": self.guidance_scale,
                "control": self.control.value if self.control else None, "state": self.state.value,
                "apply_policy": self.apply_policy.value, "estimated_wait_seconds": self.estimated_wait_seconds,
                "will_apply_at": self.will_apply_at}


@dataclass
class GenerationStatus:
    session_id: str
    state: SessionState
    current_segment: int = 0
    total_segments: int = 1
    current_block: int = 0
    total_blocks: int = 7
    frames_generated: int = 0
    gpu_memory_gb: float = 0.0
    estimated_remaining_seconds: float = 0.0
    current_prompt: str = ""
    current_guidance: float = 7.5
    queued_inputs: List[QueuedInput] = field(default_factory=list)
    message: str = ""

    @property
    def progress_percent(self) -> float:
        if self.total_segments == 0:
            return 0.0
        return (self.current_segment + self.current_block / self.total_blocks) / self.total_segments * 100


@dataclass
class CheckpointResult:
    should_continue: bool = True
    pending_input: Optional[QueuedInput] = None
    new_prompt: Optional[str] = None
    new_guidance: Optional[float] = None
    command: ControlCommand = ControlCommand.CONTINUE


@dataclass
class SegmentBoundary:
    segment_idx: int
    start_frame: int
    end_frame: int
    unique_frames: int
    overlap_with_previous: int
    is_first: bool
    is_last: bool


def calculate_total_frames(num_segments: int, segment_length: int, overlap_frames: int) -> int:
    """Unique frames of a multi-segment video: segments share `overlap_frames` with their predecessor."""
    if num_segments <= 0:
        return 0
    return num_segments * segment_length - (num_segments - 1) * overlap_frames


def validate_overlap_config(overlap_frames: int, block_size: int) -> bool:
    if overlap_frames < 0:
        raise ValueError(f"overlap_frames must be non-negative, got {overlap_frames}")
    if overlap_frames % block_size != 0:
        raise ValueError(f"overlap_frames ({overlap_frames}) must be divisible by block_size ({block_size})")
    return True


class InteractiveSession:
    def __init__(self, apply_policy: InputApplyPolicy = InputApplyPolicy.NEXT_SEGMENT, parallel_config=None):
        self.session_id = str(uuid.uuid4())[:8]
        self._apply_policy = apply_policy
        self._parallel_config = parallel_config
        self._inputs: List[QueuedInput] = []
        self._lock = threading.Lock()
        self._running = threading.Event()          # set = not paused
        self._running.set()
        self._stop = False
        self._state = SessionState.IDLE
        self._segment = self._block = 0
        self._total_segments, self._blocks_per_segment = 1, 7
        self._prompt = self._initial_prompt = ""
        self._guidance = 7.5
        self._status_cb: Optional[Callable[[GenerationStatus], None]] = None
        self._t_start = 0.0
        self._sec_per_block = 0.5
        self._blocks_done = 0

    @classmethod
    def from_prompts(cls, prompts: List[str], apply_policy: InputApplyPolicy = InputApplyPolicy.NEXT_SEGMENT,
                     parallel_config=None) -> "InteractiveSession":
        """Note the queue keeps only the LATEST unapplied input (as upstream): of several pre-queued prompts the last wins."""
        s = cls(apply_policy=apply_policy, parallel_config=parallel_config)
        if prompts:
            s.set_initial_prompt(prompts[0])
            for p in prompts[1:]:
                s.submit_input(prompt=p)
        return s

    # ---- configuration -------------------------------------------------------------------------------------------------
    def set_initial_prompt(self, prompt: str, guidance: float = 7.5):
        with self._lock:
            self._initial_prompt = self._prompt = prompt
            self._guidance = guidance

    def set_status_callback(self, callback: Callable[[GenerationStatus], None]):
        self._status_cb = callback

    def set_device(self, device):
        self._device = device

    def set_generation_params(self, total_segments: int, blocks_per_segment: int):
        with self._lock:
            self._total_segments, self._blocks_per_segment = total_segments, blocks_per_segment

    # ---- UI side -------------------------------------------------------------------------------------------------------
    def submit_input(self, prompt: Optional[str] = None, guidance_scale: Optional[float] = None,
                     control: ControlCommand = ControlCommand.CONTINUE) -> QueuedInput:
        with self._lock:
            for inp in self._inputs:
                if inp.state == InputState.QUEUED:
                    inp.state = InputState.DISCARDED
            per_seg = self._apply_policy == InputApplyPolicy.NEXT_SEGMENT
            q = QueuedInput(input_id=str(uuid.uuid4())[:8], prompt=prompt, guidance_scale=guidance_scale, control=control,
                            apply_policy=self._apply_policy, queued_at=time.time(),
                            estimated_wait_seconds=((self._blocks_per_segment - self._block) if per_seg else 1) * self._sec_per_block,
                            will_apply_at=f"Segment {self._segment + 1}" if per_seg else f"Block {self._block + 1}")
            self._inputs.append(q)
            if control == ControlCommand.PAUSE:
                self._running.clear()
            elif control == ControlCommand.RESUME:
                self._running.set()
            elif control == ControlCommand.STOP:
                self._stop = True
            return q

    # ---- generation side -----------------------------------------------------------------------------------------------
    def evaluate_checkpoint(self, checkpoint_type: str, checkpoint_index: int, current_prompt: str,
                            current_guidance: float = 7.5) -> CheckpointResult:
        with self._lock:
            if checkpoint_type == "segment":
                self._segment, self._block = checkpoint_index, 0
            else:
                self._block = checkpoint_index
            if self._stop:
                return CheckpointResult(should_continue=False, command=ControlCommand.STOP)
        while not self._running.is_set():
            if self._stop:
                return CheckpointResult(should_continue=False, command=ControlCommand.STOP)
            self._running.wait(timeout=0.1)
        with self._lock:
            mine = InputApplyPolicy.NEXT_SEGMENT if checkpoint_type == "segment" else InputApplyPolicy.NEXT_BLOCK
            if self._apply_policy != mine:
                return CheckpointResult()
            pending = next((i for i in reversed(self._inputs) if i.state == InputState.QUEUED), None)
            if pending is None:
                return CheckpointResult()
            pending.state = InputState.APPLIED
            res = CheckpointResult(pending_input=pending, command=pending.control)
            if pending.prompt:
                res.new_prompt = self._prompt = pending.prompt
            if pending.guidance_scale:
                res.new_guidance = self._guidance = pending.guidance_scale
            return res

    def update_progress(self, segment_idx: int, block_idx: int, frames_generated: int, gpu_memory_gb: float = 0.0):
        with self._lock:
            self._segment, self._block = segment_idx, block_idx
            self._state = SessionState.GENERATING
            self._blocks_done = segment_idx * self._blocks_per_segment + block_idx + 1
            if self._t_start:
                self._sec_per_block = (time.time() - self._t_start) / max(self._blocks_done, 1)
            queued = [i for i in self._inputs if i.state == InputState.QUEUED]
            left = self._total_segments * self._blocks_per_segment - self._blocks_done
            status = GenerationStatus(session_id=self.session_id, state=self._state, current_segment=self._segment,
                                      total_segments=self._total_segments, current_block=self._block,
                                      total_blocks=self._blocks_per_segment, frames_generated=frames_generated,
                                      gpu_memory_gb=gpu_memory_gb, estimated_remaining_seconds=left * self._sec_per_block,
                                      current_prompt=self._prompt, current_guidance=self._guidance, queued_inputs=queued,
                                      message=f"Generating Segment {self._segment + 1}/{self._total_segments}")
        if self._status_cb is not None:
            self._status_cb(status)

    def start_session(self):
        with self._lock:
            self._state = SessionState.GENERATING
            self._t_start = time.time()

    def end_session(self, error: bool = False):
        with self._lock:
            if self._state != SessionState.ERROR:
                self._state = SessionState.ERROR if error else SessionState.COMPLETED

    # ---- control -------------------------------------------------------------------------------------------------------
    def should_pause(self) -> bool:
        return not self._running.is_set()

    def wait_for_resume(self, timeout: Optional[float] = None):
        self._running.wait(timeout)

    def should_stop(self) -> bool:
        return self._stop

    def pause(self):
        self._running.clear()

    def resume(self):
        self._running.set()

    def stop(self):
        self._stop = True

    state = property(lambda self: self._state)
    current_prompt = property(lambda self: self._prompt)
    current_guidance = property(lambda self: self._guidance)
    initial_prompt = property(lambda self: self._initial_prompt)
    apply_policy = property(lambda self: self._apply_policy)
    is_distributed = property(lambda self: False)
