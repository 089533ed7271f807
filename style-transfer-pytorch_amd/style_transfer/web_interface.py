"""Live viewer for a running ``stylize()`` (``--web``): same public class as the reference's
``style_transfer/web_interface.py:39-148`` - ``WebInterface(host, port)``, ``put_iterate(iterate, image)``,
``put_done()``, ``close()`` - and the same HTTP surface (``/`` page, ``/image`` JPEG of the current averaged iterate,
``/websocket`` JSON messages ``{"_type": "STIterate", w, h, i, i_max, loss, time, gpu_ram}`` / ``{"_type": "WIDone"}``).

Design difference: the reference forks a server PROCESS and ships tensors through a multiprocessing queue.  Forking
after the HIP runtime is initialised is unsafe, and the only thing the server needs is a CPU copy of one small image per
iteration, so here the aiohttp application runs on a daemon THREAD with its own event loop; ``put_iterate`` hands it
the CPU tensor with ``call_soon_threadsafe``.  Cold path only.
"""

import asyncio
import dataclasses
import io
import json
import threading

INDEX_HTML = """<!doctype html>
<html><head><meta charset="utf-8"><title>style_transfer</title>
<style>body{background:#222;color:#ddd;font-family:sans-serif;margin:1em}img{max-width:100%;image-rendering:auto}
#status{margin:.5em 0;font-variant-numeric:tabular-nums}</style></head>
<body><div id="status">waiting for the first iteration...</div><img id="view" alt="">
<script>
let last = null, rate = null, pending = false;
const view = document.getElementById('view'), status = document.getElementById('status');
function refresh() { if (pending) return; pending = true; const im = new Image();
  im.onload = () => { view.src = im.src; pending = false; }; im.onerror = () => { pending = false; };
  im.src = '/image?' + Date.now(); }
const ws = new WebSocket((location.protocol === 'https:' ? 'wss://' : 'ws://') + location.host + '/websocket');
ws.onmessage = (ev) => { const m = JSON.parse(ev.data);
  if (m._type === 'STIterate') {
    if (last !== null && m.time > last) { const r = 1 / (m.time - last); rate = rate === null ? r : 0.9 * rate + 0.1 * r; }
    last = m.time;
    status.textContent = `Size: ${m.w}x${m.h}, iteration ${m.i} / ${m.i_max}, loss ${m.loss.toPrecision(6)}` +
      (rate ? `, ${rate.toFixed(2)} iterations per second` : '') + `, GPU RAM ${(m.gpu_ram / 2 ** 20).toFixed(0)} MB`;
    refresh();
  } else if (m._type === 'WIDone') { status.textContent += ' - done'; refresh(); } };
ws.onclose = () => { status.textContent += ' (connection closed)'; };
</script></body></html>
"""


@dataclasses.dataclass
class WIDone:
    pass


def _encode(msg):
    fields = dict(msg.__dict__) if dataclasses.is_dataclass(msg) else dict(msg)
    fields['_type'] = type(msg).__name__
    return json.dumps(fields)


class WebInterface:
    def __init__(self, host, port):
        from aiohttp import web
        self.host, self.port = host, port
        self._web = web
        self._image = None            # CPU float tensor, 3 x H x W in [0, 1]
        self._sockets = []
        self._loop = asyncio.new_event_loop()
        self._runner = None
        self._ready = threading.Event()
        self._error = None
        print(f'Starting web interface at http://{self.host}:{self.port}/')
        self._thread = threading.Thread(target=self._serve, name='style_transfer-web', daemon=True)
        self._thread.start()
        self._ready.wait(10)
        if self._error is not None:
            raise self._error

    # ---- server thread ----
    def _serve(self):
        asyncio.set_event_loop(self._loop)
        try:
            self._loop.run_until_complete(self._start())
        except Exception as err:            # noqa: BLE001 - surfaced to the constructor
            self._error = err
            self._ready.set()
            return
        self._ready.set()
        self._loop.run_forever()
        self._loop.run_until_complete(self._stop())
        self._loop.close()

    async def _start(self):
        web = self._web
        app = web.Application()
        app.router.add_routes([web.get('/', self._index), web.get('/image', self._jpeg),
                               web.get('/websocket', self._websocket)])
        self._runner = web.AppRunner(app)
        await self._runner.setup()
        await web.TCPSite(self._runner, self.host, self.port).start()

    async def _stop(self):
        for ws in list(self._sockets):
            await ws.close()
        if self._runner is not None:
            await self._runner.cleanup()

    async def _index(self, request):
        return self._web.Response(text=INDEX_HTML, content_type='text/html')

    def _compress(self):
        from . import srgb_profile
        from .style_transfer import to_pil_image
        buf = io.BytesIO()
        to_pil_image(self._image).save(buf, format='jpeg', icc_profile=srgb_profile, quality=95, subsampling=0)
        return buf.getvalue()

    async def _jpeg(self, request):
        if self._image is None:
            raise self._web.HTTPNotFound()
        body = await self._loop.run_in_executor(None, self._compress)
        return self._web.Response(body=body, content_type='image/jpeg')

    async def _websocket(self, request):
        ws = self._web.WebSocketResponse()
        await ws.prepare(request)
        self._sockets.append(ws)
        try:
            async for _ in ws:
                pass
        finally:
            if ws in self._sockets:
                self._sockets.remove(ws)
        return ws

    async def _broadcast(self, text):
        for ws in list(self._sockets):
            try:
                await ws.send_str(text)
            except (ConnectionError, RuntimeError):
                if ws in self._sockets:
                    self._sockets.remove(ws)

    def _post(self, image, text):
        if image is not None:
            self._image = image
        asyncio.ensure_future(self._broadcast(text))

    # ---- called from the stylize() thread ----
    def put_iterate(self, iterate, image):
        self._loop.call_soon_threadsafe(self._post, image.detach().cpu(), _encode(iterate))

    def put_done(self):
        self._loop.call_soon_threadsafe(self._post, None, _encode(WIDone()))

    def close(self):
        if self._thread.is_alive():
            self._loop.call_soon_threadsafe(self._loop.stop)
            self._thread.join(12)
